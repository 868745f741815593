// sdl.cpp — scene description parsers.  Grammar and semantics follow the reference's
// src/sdl/scene_parser.cpp:71-451 (text) and scene_parser_json.cpp:27-195 (JSON); see sdl.h.
#include "sdl.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <mutex>
#include <sstream>

namespace lr {

namespace fs = std::filesystem;

static int g_log_level = 1;
void set_log_level(int level) { g_log_level = level; }
void log_info(const std::string &msg) {
    if (g_log_level >= 2) { std::fprintf(stderr, "[info] %s\n", msg.c_str()); }
}
void log_warning(const std::string &msg) {
    if (g_log_level >= 1) { std::fprintf(stderr, "[warning] %s\n", msg.c_str()); }
}

const char *tag_description(Tag tag) {
    switch (tag) {
        case Tag::ROOT: return "__root__";
        case Tag::INTERNAL: return "__internal__";
        case Tag::CAMERA: return "Camera";
        case Tag::SHAPE: return "Shape";
        case Tag::SURFACE: return "Surface";
        case Tag::LIGHT: return "Light";
        case Tag::TRANSFORM: return "Transform";
        case Tag::FILM: return "Film";
        case Tag::FILTER: return "Filter";
        case Tag::SAMPLER: return "Sampler";
        case Tag::INTEGRATOR: return "Integrator";
        case Tag::LIGHT_SAMPLER: return "LightSampler";
        case Tag::ENVIRONMENT: return "Environment";
        case Tag::TEXTURE: return "Texture";
        case Tag::TEXTURE_MAPPING: return "TextureMapping";
        case Tag::SPECTRUM: return "Spectrum";
        case Tag::MEDIUM: return "Medium";
        case Tag::PHASE_FUNCTION: return "PhaseFunction";
        case Tag::DECLARATION: return "__declaration__";
    }
    return "__invalid__";
}

Tag parse_tag(std::string t) {
    for (auto &c : t) { c = static_cast<char>(std::tolower(c)); }
    static const std::unordered_map<std::string, Tag> table{
        {"camera", Tag::CAMERA}, {"cam", Tag::CAMERA},
        {"shape", Tag::SHAPE}, {"object", Tag::SHAPE}, {"obj", Tag::SHAPE},
        {"surface", Tag::SURFACE}, {"surf", Tag::SURFACE},
        {"lightsource", Tag::LIGHT}, {"light", Tag::LIGHT}, {"illuminant", Tag::LIGHT}, {"illum", Tag::LIGHT},
        {"transform", Tag::TRANSFORM}, {"xform", Tag::TRANSFORM},
        {"film", Tag::FILM}, {"filter", Tag::FILTER}, {"sampler", Tag::SAMPLER},
        {"integrator", Tag::INTEGRATOR}, {"lightsampler", Tag::LIGHT_SAMPLER},
        {"environment", Tag::ENVIRONMENT}, {"env", Tag::ENVIRONMENT},
        {"texture", Tag::TEXTURE}, {"tex", Tag::TEXTURE},
        {"texturemapping", Tag::TEXTURE_MAPPING}, {"texmapping", Tag::TEXTURE_MAPPING},
        {"spectrum", Tag::SPECTRUM}, {"spec", Tag::SPECTRUM},
        {"generic", Tag::DECLARATION}, {"template", Tag::DECLARATION},
        {"medium", Tag::MEDIUM}, {"phasefunction", Tag::PHASE_FUNCTION}};
    auto it = table.find(t);
    return it == table.end() ? Tag::ROOT : it->second;
}

// ---------------------------------------------------------------- NodeDesc

std::string NodeDesc::location() const {
    return _source_file.empty() ? std::string{"<unknown>"} : _source_file + ":" + std::to_string(_line);
}

void NodeDesc::define(Tag tag, const std::string &impl, const std::string &file, uint32_t line,
                      const NodeDesc *base) {
    _tag = tag;
    _impl = impl;
    for (auto &c : _impl) { c = static_cast<char>(std::tolower(c)); }
    _source_file = file;
    _line = line;
    _base = base;
}

NodeDesc *NodeDesc::define_internal(const std::string &impl, const std::string &file, uint32_t line,
                                    const NodeDesc *base) {
    auto id = _identifier + ".$internal" + std::to_string(_internal.size());
    auto node = std::make_unique<NodeDesc>(id, Tag::INTERNAL);
    node->define(Tag::INTERNAL, impl, file, line, base);
    _internal.emplace_back(std::move(node));
    return _internal.back().get();
}

void NodeDesc::add_property(const std::string &name, value_list v) {
    if (!_props.emplace(name, std::move(v)).second) {
        throw Error{"Redefinition of property '" + name + "' in scene description node '" +
                    _identifier + "'. [" + location() + "]"};
    }
}

const NodeDesc::value_list *NodeDesc::_find(const std::string &name) const {
    auto it = _props.find(name);
    if (it != _props.end()) { return &it->second; }
    return _base ? _base->_find(name) : nullptr;
}

bool NodeDesc::has_property(const std::string &name) const { return _find(name) != nullptr; }
bool NodeDesc::is_string_property(const std::string &name) const {
    auto v = _find(name);
    return v != nullptr && std::holds_alternative<string_list>(*v);
}

template<typename L>
const L *NodeDesc::_raw(const std::string &name) const {
    auto v = _find(name);
    if (v == nullptr) { return nullptr; }
    auto p = std::get_if<L>(v);
    if (p == nullptr) {
        log_warning("Property '" + name + "' is defined but has an unexpected list type in node '" +
                    _identifier + "'. [" + location() + "]");
    }
    return p;
}

std::optional<double> NodeDesc::number_opt(const std::string &name) const {
    auto p = _raw<number_list>(name);
    if (p == nullptr || p->empty()) { return std::nullopt; }
    if (p->size() > 1u) {
        log_warning("Found " + std::to_string(p->size()) + " values for property '" + name +
                    "' in node '" + _identifier + "', only 1 is required.");
    }
    return p->front();
}
std::optional<std::vector<double>> NodeDesc::numbers_opt(const std::string &name) const {
    auto p = _raw<number_list>(name);
    if (p == nullptr) { return std::nullopt; }
    return *p;
}
std::optional<std::vector<double>> NodeDesc::vector_opt(const std::string &name, size_t n) const {
    auto p = _raw<number_list>(name);
    if (p == nullptr || p->empty()) { return std::nullopt; }
    if (p->size() < n) {
        log_warning("Required " + std::to_string(n) + " values but found " + std::to_string(p->size()) +
                    " for property '" + name + "' in node '" + _identifier + "'.");
        return std::nullopt;
    }
    return std::vector<double>(p->begin(), p->begin() + static_cast<long>(n));
}
std::optional<bool> NodeDesc::bool_opt(const std::string &name) const {
    auto p = _raw<bool_list>(name);
    if (p == nullptr || p->empty()) { return std::nullopt; }
    return static_cast<bool>(p->front());
}
std::optional<std::string> NodeDesc::string_opt(const std::string &name) const {
    auto p = _raw<string_list>(name);
    if (p == nullptr || p->empty()) { return std::nullopt; }
    return p->front();
}
std::optional<std::vector<std::string>> NodeDesc::strings_opt(const std::string &name) const {
    auto p = _raw<string_list>(name);
    if (p == nullptr) { return std::nullopt; }
    return *p;
}
const NodeDesc *NodeDesc::node_or_null(const std::string &name) const {
    auto p = _raw<node_list>(name);
    if (p == nullptr || p->empty()) { return nullptr; }
    return p->front();
}
std::optional<NodeDesc::node_list> NodeDesc::nodes_opt(const std::string &name) const {
    auto p = _raw<node_list>(name);
    if (p == nullptr) { return std::nullopt; }
    return *p;
}

float NodeDesc::float_or(const std::string &name, float dv) const {
    auto v = number_opt(name);
    return v ? static_cast<float>(*v) : dv;
}
uint32_t NodeDesc::uint_or(const std::string &name, uint32_t dv) const {
    auto v = number_opt(name);
    if (!v) { return dv; }
    auto u = static_cast<uint32_t>(*v);
    if (static_cast<double>(u) != *v) {
        throw Error{"Cannot convert property '" + name + "' to integer in node '" + _identifier +
                    "'. [" + location() + "]"};
    }
    return u;
}
bool NodeDesc::bool_or(const std::string &name, bool dv) const { return bool_opt(name).value_or(dv); }
std::string NodeDesc::string_or(const std::string &name, const std::string &dv) const {
    return string_opt(name).value_or(dv);
}
std::string NodeDesc::path_or(const std::string &name, const std::string &dv) const {
    auto s = string_opt(name);
    if (!s) { return dv; }
    fs::path p{*s};
    if (_source_file.empty() || p.is_absolute()) { return p.string(); }
    return (fs::path{_source_file}.parent_path() / p).lexically_normal().string();
}
std::vector<float> NodeDesc::float_list_or_empty(const std::string &name) const {
    std::vector<float> out;
    if (auto p = _raw<number_list>(name)) {
        out.reserve(p->size());
        for (auto v : *p) { out.emplace_back(static_cast<float>(v)); }
    }
    return out;
}
std::vector<float> NodeDesc::float_list(const std::string &name) const {
    if (_raw<number_list>(name) == nullptr) {
        throw Error{"No valid values given for property '" + name + "' in node '" + _identifier +
                    "'. [" + location() + "]"};
    }
    return float_list_or_empty(name);
}
std::vector<uint32_t> NodeDesc::uint_list(const std::string &name) const {
    auto p = _raw<number_list>(name);
    if (p == nullptr) {
        throw Error{"No valid values given for property '" + name + "' in node '" + _identifier +
                    "'. [" + location() + "]"};
    }
    std::vector<uint32_t> out;
    out.reserve(p->size());
    for (auto v : *p) {
        auto u = static_cast<uint32_t>(v);
        if (static_cast<double>(u) != v) {
            throw Error{"Non-integral value in property '" + name + "' of node '" + _identifier + "'."};
        }
        out.emplace_back(u);
    }
    return out;
}
const NodeDesc *NodeDesc::node(const std::string &name) const {
    auto n = node_or_null(name);
    if (n == nullptr) {
        throw Error{"No valid values given for property '" + name + "' in node '" + _identifier +
                    "'. [" + location() + "]"};
    }
    return n;
}
NodeDesc::node_list NodeDesc::node_list_required(const std::string &name) const {
    auto n = nodes_opt(name);
    if (!n) {
        throw Error{"No valid values given for property '" + name + "' in node '" + _identifier +
                    "'. [" + location() + "]"};
    }
    return *n;
}
NodeDesc::node_list NodeDesc::node_list_or_empty(const std::string &name) const {
    return nodes_opt(name).value_or(node_list{});
}

const NodeDesc *NodeDesc::shared_default(Tag tag, std::string impl) {
    static std::unordered_map<std::string, std::unique_ptr<NodeDesc>> descs;
    static std::mutex mutex;
    for (auto &c : impl) { c = static_cast<char>(std::tolower(c)); }
    auto id = std::string{"__shared_default_"} + tag_description(tag) + "_" + impl;
    for (auto &c : id) { c = static_cast<char>(std::tolower(c)); }
    std::scoped_lock lock{mutex};
    auto it = descs.find(id);
    if (it != descs.end()) { return it->second.get(); }
    auto d = std::make_unique<NodeDesc>(id, tag);
    d->define(tag, impl, {}, 0u);
    return descs.emplace(id, std::move(d)).first->second.get();
}

// ---------------------------------------------------------------- SceneDesc

const NodeDesc *SceneDesc::node(const std::string &id) const {
    auto it = _globals.find(id);
    return it == _globals.end() ? nullptr : it->second.get();
}

const NodeDesc *SceneDesc::reference(const std::string &id) {
    auto it = _globals.find(id);
    if (it != _globals.end()) { return it->second.get(); }
    auto d = std::make_unique<NodeDesc>(id, Tag::DECLARATION);
    return _globals.emplace(id, std::move(d)).first->second.get();
}

NodeDesc *SceneDesc::define(const std::string &id, Tag tag, const std::string &impl,
                            const std::string &file, uint32_t line, const NodeDesc *base) {
    if (id == root_node_identifier || tag == Tag::ROOT || tag == Tag::INTERNAL) {
        throw Error{"Defining internal or root node as a global node is not allowed. [" + file + ":" +
                    std::to_string(line) + "]"};
    }
    auto it = _globals.find(id);
    if (it == _globals.end()) {
        it = _globals.emplace(id, std::make_unique<NodeDesc>(id, tag)).first;
    }
    auto n = it->second.get();
    if (n->is_defined()) {
        throw Error{"Redefinition of node '" + id + "' (" + n->location() + ") at " + file + ":" +
                    std::to_string(line) + "."};
    }
    n->define(tag, impl, file, line, base);
    return n;
}

NodeDesc *SceneDesc::define_root(const std::string &file, uint32_t line) {
    if (_root.is_defined()) {
        throw Error{"Redefinition of root node at " + file + ":" + std::to_string(line) + "."};
    }
    _root.define(Tag::ROOT, root_node_identifier, file, line);
    return &_root;
}

const std::string &SceneDesc::register_file(const std::string &path) {
    _files.emplace_back(path);
    return _files.back();
}

void SceneDesc::validate() const {
    for (auto &[id, n] : _globals) {
        if (!n->is_defined() && n->tag() == Tag::DECLARATION && n->impl_type().empty()) {
            // pure templates ("generic x {...}") never get an impl; only flag undefined references
            if (n->properties().empty()) {
                throw Error{"Node '" + id + "' is referenced but never defined."};
            }
        }
    }
}

// ---------------------------------------------------------------- text parser

namespace {

void dispatch_parse(SceneDesc &desc, const fs::path &path, const MacroMap &cli_macros);

class TextParser {
    SceneDesc &_desc;
    const MacroMap &_cli_macros;
    MacroMap _local_macros;
    std::string _file;
    std::string _source;
    size_t _cursor{0};
    uint32_t _line{1}, _column{0};
    std::vector<std::string> _parsing_macros;

    [[noreturn]] void _error(const std::string &msg) const {
        throw Error{msg + " [" + _file + ":" + std::to_string(_line) + ":" + std::to_string(_column) + "]"};
    }
    bool _eof() const { return _parsing_macros.empty() && _cursor >= _source.size(); }

    char _peek_char() {
        if (!_parsing_macros.empty()) { return _parsing_macros.back().front(); }
        if (_eof()) { _error("Premature EOF."); }
        auto c = _source[_cursor];
        if (c == '\r') {
            if (_cursor + 1u < _source.size() && _source[_cursor + 1u] == '\n') { _cursor++; }
            return '\n';
        }
        return c;
    }
    char _get_char() {
        if (!_parsing_macros.empty()) {
            auto m = _parsing_macros.back();
            _parsing_macros.pop_back();
            auto c = m.front();
            if (m.size() > 1u) { _parsing_macros.emplace_back(m.substr(1u)); }
            return c;
        }
        if (_eof()) { _error("Premature EOF."); }
        auto c = _source[_cursor++];
        if (c == '\r') {
            if (_cursor < _source.size() && _source[_cursor] == '\n') { _cursor++; }
            _line++, _column = 0;
            return '\n';
        }
        if (c == '\n') { _line++, _column = 0; } else { _column++; }
        return c;
    }
    char _peek(bool escape_macro = false) {
        auto c = _peek_char();
        if (!escape_macro) {
            while (c == '#') {
                (void)_get_char();
                _parse_macro();
                c = _peek_char();
            }
        }
        return c;
    }
    char _get(bool escape_macro = false) {
        auto c = _get_char();
        if (!escape_macro) {
            while (c == '#') {
                _parse_macro();
                c = _get_char();
            }
        }
        return c;
    }
    void _skip() { (void)_get(true); }
    void _match(char c) {
        if (auto got = _get(); got != c) {
            _error(std::string{"Invalid character '"} + got + "' (expected '" + c + "').");
        }
    }
    void _skip_blanks() {
        while (!_eof()) {
            auto c = _peek(true);
            if (c == ' ' || c == '\t' || c == '\n') {
                _skip();
            } else if (c == '/') {
                _skip();
                _match('/');
                while (!_eof() && _get(true) != '\n') {}
            } else {
                break;
            }
        }
    }
    std::string _read_identifier(bool escape_macro = false) {
        std::string id;
        auto c = _get(escape_macro);
        if (c != '$' && c != '_' && !std::isalpha(static_cast<unsigned char>(c))) {
            _error(std::string{"Invalid character '"} + c + "' in identifier.");
        }
        id.push_back(c);
        auto body = [](char ch) {
            return std::isalnum(static_cast<unsigned char>(ch)) || ch == '_' || ch == '$' || ch == '-';
        };
        while (!_eof() && body(_peek(escape_macro))) { id.push_back(_get(escape_macro)); }
        return id;
    }
    double _read_number() {
        std::string s;
        if (auto c = _peek(); c == '+') {
            _skip();
            _skip_blanks();
        } else if (c == '-') {
            s.push_back(_get());
            _skip_blanks();
        }
        auto is_digit = [](char ch) {
            return std::isdigit(static_cast<unsigned char>(ch)) || ch == '.' || ch == 'e' || ch == '-' || ch == '+';
        };
        while (!_eof() && is_digit(_peek())) { s.push_back(_get()); }
        char *end = nullptr;
        auto value = std::strtod(s.c_str(), &end);
        if (s.empty() || end != s.c_str() + s.size()) {
            _error("Invalid number string '" + s.substr(0, 8) + "...'.");
        }
        return value;
    }
    bool _read_bool() {
        if (_peek() == 't') {
            for (auto x : std::string{"true"}) { _match(x); }
            return true;
        }
        for (auto x : std::string{"false"}) { _match(x); }
        return false;
    }
    std::string _read_string() {
        auto quote = _get();
        if (quote != '"' && quote != '\'') { _error(std::string{"Expected string but got "} + quote + "."); }
        std::string s;
        for (auto c = _get(); c != quote; c = _get()) {
            if (!std::isprint(static_cast<unsigned char>(c))) { _error("Unexpected non-printable character."); }
            if (c == '\\') {
                auto esc = _get(true);
                switch (esc) {
                    case 'b': c = '\b'; break;
                    case 'f': c = '\f'; break;
                    case 'n': c = '\n'; break;
                    case 'r': c = '\r'; break;
                    case 't': c = '\t'; break;
                    case '\\': c = '\\'; break;
                    case '\'': c = '\''; break;
                    case '"': c = '"'; break;
                    case '#': c = '#'; break;
                    default: _error(std::string{"Invalid escaped character '"} + esc + "'.");
                }
            }
            s.push_back(c);
        }
        return s;
    }
    void _parse_macro() {
        _skip_blanks();
        auto key = _read_identifier(true);
        if (auto it = _cli_macros.find(key); it != _cli_macros.end()) {
            if (!it->second.empty()) { _parsing_macros.emplace_back(it->second); }
        } else if (auto lt = _local_macros.find(key); lt != _local_macros.end()) {
            if (!lt->second.empty()) { _parsing_macros.emplace_back(lt->second); }
        } else {
            _error("Undefined macro '" + key + "'.");
        }
    }
    void _parse_define() {
        _skip_blanks();
        auto key = _read_identifier(true);
        _skip_blanks();
        std::string value;
        while (!_eof() && _peek(true) != '\n' && _peek(true) != '/') { value.push_back(_get(true)); }
        if (_cli_macros.count(key)) {
            log_warning("Local macro '" + key + "' is shadowed by command-line definition.");
        } else {
            auto inserted = _local_macros.insert_or_assign(key, value).second;
            if (!inserted) { log_warning("Macro '" + key + "' is redefined."); }
        }
    }
    const NodeDesc *_parse_base_node() {
        _match('(');
        _skip_blanks();
        _match('@');
        _skip_blanks();
        auto base = _desc.reference(_read_identifier());
        _skip_blanks();
        _match(')');
        return base;
    }
    NodeDesc::value_list _parse_value_list(NodeDesc *node) {
        _match('{');
        _skip_blanks();
        NodeDesc::value_list result;
        auto c = _peek();
        if (c == '}') { _error("Empty value list."); }
        if (c == '@' || std::isupper(static_cast<unsigned char>(c))) {
            NodeDesc::node_list list;
            auto ref_or_def = [&]() -> const NodeDesc * {
                if (_peek() == '@') {
                    _skip();
                    _skip_blanks();
                    return _desc.reference(_read_identifier());
                }
                auto line = _line;
                auto impl = _read_identifier();
                const NodeDesc *base = nullptr;
                if (_peek() == '(') { base = _parse_base_node(); }
                auto internal = node->define_internal(impl, _file, line, base);
                _parse_node_body(internal);
                return internal;
            };
            list.emplace_back(ref_or_def());
            _skip_blanks();
            while (_peek() != '}') {
                _match(',');
                _skip_blanks();
                list.emplace_back(ref_or_def());
                _skip_blanks();
            }
            result = std::move(list);
        } else if (c == '"' || c == '\'') {
            NodeDesc::string_list list;
            list.emplace_back(_read_string());
            _skip_blanks();
            while (_peek() != '}') {
                _match(',');
                _skip_blanks();
                list.emplace_back(_read_string());
                _skip_blanks();
            }
            result = std::move(list);
        } else if (c == 't' || c == 'f') {
            NodeDesc::bool_list list;
            list.emplace_back(_read_bool());
            _skip_blanks();
            while (_peek() != '}') {
                _match(',');
                _skip_blanks();
                list.emplace_back(_read_bool());
                _skip_blanks();
            }
            result = std::move(list);
        } else {
            NodeDesc::number_list list;
            list.emplace_back(_read_number());
            _skip_blanks();
            while (_peek() != '}') {
                _match(',');
                _skip_blanks();
                list.emplace_back(_read_number());
                _skip_blanks();
            }
            result = std::move(list);
        }
        _skip_blanks();
        _match('}');
        return result;
    }
    void _parse_node_body(NodeDesc *node) {
        _skip_blanks();
        _match('{');
        _skip_blanks();
        while (_peek() != '}') {
            auto prop = _read_identifier();
            _skip_blanks();
            if (_peek() == ':') {// inline node
                _skip();
                _skip_blanks();
                auto line = _line;
                auto impl = _read_identifier();
                const NodeDesc *base = nullptr;
                if (_peek() == '(') { base = _parse_base_node(); }
                auto internal = node->define_internal(impl, _file, line, base);
                _parse_node_body(internal);
                node->add_property(prop, NodeDesc::node_list{internal});
            } else {
                node->add_property(prop, _parse_value_list(node));
            }
            _skip_blanks();
        }
        _match('}');
    }
    void _parse_global_node(uint32_t line, const std::string &tag_desc) {
        auto tag = parse_tag(tag_desc);
        if (tag == Tag::ROOT) { _error("Invalid scene node type '" + tag_desc + "'."); }
        _skip_blanks();
        auto name = _read_identifier();
        _skip_blanks();
        const NodeDesc *base = nullptr;
        std::string impl;
        if (_peek() == ':') {
            _match(':');
            _skip_blanks();
            impl = _read_identifier();
            _skip_blanks();
            if (_peek() == '(') { base = _parse_base_node(); }
            _skip_blanks();
        }
        _parse_node_body(_desc.define(name, tag, impl, _file, line, base));
    }

public:
    TextParser(SceneDesc &desc, std::string file, std::string source, const MacroMap &cli)
        : _desc{desc}, _cli_macros{cli}, _file{std::move(file)}, _source{std::move(source)} {}

    void parse() {
        _skip_blanks();
        while (!_eof()) {
            auto line = _line;
            auto token = _read_identifier();
            if (token == "import") {
                _skip_blanks();
                fs::path path{_read_string()};
                if (!path.is_absolute()) { path = fs::path{_file}.parent_path() / path; }
                dispatch_parse(_desc, path, _cli_macros);
            } else if (token == "define") {
                _parse_define();
            } else if (token == SceneDesc::root_node_identifier) {
                _parse_node_body(_desc.define_root(_file, line));
            } else {
                _parse_global_node(line, token);
            }
            _skip_blanks();
        }
    }
};

// ---------------------------------------------------------------- minimal JSON

struct Json {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind{NUL};
    bool b{false};
    double num{0.};
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;// insertion-ordered
    const Json *find(const std::string &k) const {
        for (auto &kv : obj) {
            if (kv.first == k) { return &kv.second; }
        }
        return nullptr;
    }
};

class JsonReader {
    const std::string &_s;
    size_t _p{0};
    const std::string &_file;
    [[noreturn]] void _error(const std::string &m) const {
        throw Error{"JSON: " + m + " at offset " + std::to_string(_p) + " [" + _file + "]"};
    }
    void _ws() {
        while (_p < _s.size()) {
            auto c = _s[_p];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                _p++;
            } else if (c == '/' && _p + 1 < _s.size() && _s[_p + 1] == '/') {
                while (_p < _s.size() && _s[_p] != '\n') { _p++; }
            } else {
                break;
            }
        }
    }
    std::string _string() {
        if (_s[_p] != '"') { _error("expected string"); }
        _p++;
        std::string out;
        while (_p < _s.size() && _s[_p] != '"') {
            auto c = _s[_p++];
            if (c == '\\') {
                if (_p >= _s.size()) { _error("bad escape"); }
                auto e = _s[_p++];
                switch (e) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        if (_p + 4 > _s.size()) { _error("bad \\u escape"); }
                        auto code = std::strtoul(_s.substr(_p, 4).c_str(), nullptr, 16);
                        _p += 4;
                        if (code < 0x80) {
                            out.push_back(static_cast<char>(code));
                        } else if (code < 0x800) {
                            out.push_back(static_cast<char>(0xc0 | (code >> 6)));
                            out.push_back(static_cast<char>(0x80 | (code & 0x3f)));
                        } else {
                            out.push_back(static_cast<char>(0xe0 | (code >> 12)));
                            out.push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3f)));
                            out.push_back(static_cast<char>(0x80 | (code & 0x3f)));
                        }
                        break;
                    }
                    default: out.push_back(e); break;
                }
            } else {
                out.push_back(c);
            }
        }
        if (_p >= _s.size()) { _error("unterminated string"); }
        _p++;
        return out;
    }

public:
    JsonReader(const std::string &s, const std::string &file) : _s{s}, _file{file} {}
    Json value() {
        _ws();
        if (_p >= _s.size()) { _error("unexpected end"); }
        Json j;
        auto c = _s[_p];
        if (c == '{') {
            j.kind = Json::OBJ;
            _p++;
            _ws();
            if (_p < _s.size() && _s[_p] == '}') { _p++; return j; }
            for (;;) {
                _ws();
                auto k = _string();
                _ws();
                if (_p >= _s.size() || _s[_p] != ':') { _error("expected ':'"); }
                _p++;
                j.obj.emplace_back(std::move(k), value());
                _ws();
                if (_p < _s.size() && _s[_p] == ',') { _p++; continue; }
                if (_p < _s.size() && _s[_p] == '}') { _p++; break; }
                _error("expected ',' or '}'");
            }
        } else if (c == '[') {
            j.kind = Json::ARR;
            _p++;
            _ws();
            if (_p < _s.size() && _s[_p] == ']') { _p++; return j; }
            for (;;) {
                j.arr.emplace_back(value());
                _ws();
                if (_p < _s.size() && _s[_p] == ',') { _p++; continue; }
                if (_p < _s.size() && _s[_p] == ']') { _p++; break; }
                _error("expected ',' or ']'");
            }
        } else if (c == '"') {
            j.kind = Json::STR;
            j.str = _string();
        } else if (_s.compare(_p, 4, "true") == 0) {
            j.kind = Json::BOOL, j.b = true, _p += 4;
        } else if (_s.compare(_p, 5, "false") == 0) {
            j.kind = Json::BOOL, j.b = false, _p += 5;
        } else if (_s.compare(_p, 4, "null") == 0) {
            j.kind = Json::NUL, _p += 4;
        } else {
            char *end = nullptr;
            j.kind = Json::NUM;
            j.num = std::strtod(_s.c_str() + _p, &end);
            if (end == _s.c_str() + _p) { _error("invalid token"); }
            _p = static_cast<size_t>(end - _s.c_str());
        }
        return j;
    }
    void finish() {
        _ws();
        if (_p != _s.size()) { _error("trailing characters"); }
    }
};

class JsonSceneParser {
    SceneDesc &_desc;
    const MacroMap &_cli_macros;
    std::string _file;

    const NodeDesc *_reference(const std::string &name) const {
        if (name.empty() || name[0] != '@') { throw Error{"Invalid reference name '" + name + "'."}; }
        return _desc.reference(name.substr(1));
    }
    const NodeDesc *_parse_internal(NodeDesc &desc, const std::string &key, const Json &n) const {
        if (n.kind != Json::OBJ) { throw Error{"Invalid internal node '" + key + "' in " + _file}; }
        for (auto &kv : n.obj) {
            if (kv.first != "impl" && kv.first != "base" && kv.first != "prop") {
                throw Error{"Invalid internal node property '" + key + "." + kv.first + "'."};
            }
        }
        auto impl = n.find("impl");
        if (impl == nullptr || impl->kind != Json::STR) { throw Error{"Missing impl in internal node '" + key + "'."}; }
        const NodeDesc *base = nullptr;
        if (auto b = n.find("base")) { base = _reference(b->str); }
        auto internal = desc.define_internal(impl->str, _file, 0u, base);
        if (auto p = n.find("prop")) { _parse_node(*internal, *p); }
        return internal;
    }
    void _parse_node(NodeDesc &desc, const Json &node) const {
        if (node.kind != Json::OBJ) { throw Error{"Invalid node body for '" + desc.identifier() + "'."}; }
        for (auto &[key, v] : node.obj) {
            switch (v.kind) {
                case Json::STR:
                    if (!v.str.empty() && v.str[0] == '@') {
                        desc.add_property(key, NodeDesc::node_list{_reference(v.str)});
                    } else {
                        desc.add_property(key, NodeDesc::string_list{v.str});
                    }
                    break;
                case Json::NUM: desc.add_property(key, NodeDesc::number_list{v.num}); break;
                case Json::BOOL: desc.add_property(key, NodeDesc::bool_list{v.b}); break;
                case Json::ARR: {
                    if (v.arr.empty()) { throw Error{"Empty array is not allowed in '" + desc.identifier() + "'.'" + key + "'."}; }
                    auto &first = v.arr[0];
                    if (first.kind == Json::STR && !(!first.str.empty() && first.str[0] == '@')) {
                        NodeDesc::string_list l;
                        for (auto &e : v.arr) { l.emplace_back(e.str); }
                        desc.add_property(key, std::move(l));
                    } else if (first.kind == Json::NUM) {
                        NodeDesc::number_list l;
                        for (auto &e : v.arr) { l.emplace_back(e.num); }
                        desc.add_property(key, std::move(l));
                    } else if (first.kind == Json::BOOL) {
                        NodeDesc::bool_list l;
                        for (auto &e : v.arr) { l.emplace_back(e.b); }
                        desc.add_property(key, std::move(l));
                    } else {
                        NodeDesc::node_list l;
                        for (auto &e : v.arr) {
                            l.emplace_back(e.kind == Json::STR ? _reference(e.str) : _parse_internal(desc, key, e));
                        }
                        desc.add_property(key, std::move(l));
                    }
                    break;
                }
                case Json::OBJ: desc.add_property(key, NodeDesc::node_list{_parse_internal(desc, key, v)}); break;
                case Json::NUL: break;
            }
        }
    }

public:
    JsonSceneParser(SceneDesc &desc, std::string file, const MacroMap &cli)
        : _desc{desc}, _cli_macros{cli}, _file{std::move(file)} {}

    void parse(const std::string &source) {
        JsonReader reader{source, _file};
        auto root = reader.value();
        reader.finish();
        if (root.kind != Json::OBJ) { throw Error{"JSON scene root must be an object. [" + _file + "]"}; }
        if (auto imp = root.find("import")) {
            auto one = [&](const Json &j) {
                if (j.kind != Json::STR) { throw Error{"Invalid import node. [" + _file + "]"}; }
                fs::path p{j.str};
                if (!p.is_absolute()) { p = fs::path{_file}.parent_path() / p; }
                dispatch_parse(_desc, p, _cli_macros);
            };
            if (imp->kind == Json::ARR) {
                for (auto &e : imp->arr) { one(e); }
            } else {
                one(*imp);
            }
        }
        for (auto &[key, v] : root.obj) {
            if (key == SceneDesc::root_node_identifier) {
                _parse_node(*_desc.define_root(_file, 0u), v);
            } else if (key != "import") {
                if (v.kind != Json::OBJ) { throw Error{"Invalid global node '" + key + "'."}; }
                for (auto &kv : v.obj) {
                    if (kv.first != "type" && kv.first != "impl" && kv.first != "base" && kv.first != "prop") {
                        throw Error{"Invalid global node property '" + key + "." + kv.first + "'."};
                    }
                }
                auto type = v.find("type");
                auto impl = v.find("impl");
                if (type == nullptr || type->kind != Json::STR) { throw Error{"Missing node type in global node '" + key + "'."}; }
                if (impl == nullptr || impl->kind != Json::STR) { throw Error{"Missing impl in global node '" + key + "'."}; }
                auto tag = parse_tag(type->str);
                if (tag == Tag::ROOT) { throw Error{"Unknown scene node type: " + type->str}; }
                const NodeDesc *base = nullptr;
                if (auto b = v.find("base")) { base = _reference(b->str); }
                auto global = _desc.define(key, tag, impl->str, _file, 0u, base);
                if (auto p = v.find("prop")) { _parse_node(*global, *p); }
            }
        }
    }
};

bool has_json_extension(const fs::path &p) {
    auto ext = p.extension().string();
    for (auto &c : ext) { c = static_cast<char>(std::tolower(c)); }
    return ext == ".json";
}

void dispatch_parse(SceneDesc &desc, const fs::path &path, const MacroMap &cli_macros) {
    std::ifstream file{path, std::ios::binary};
    if (!file) { throw Error{"Failed to open scene file '" + path.string() + "'."}; }
    std::string source{std::istreambuf_iterator<char>{file}, std::istreambuf_iterator<char>{}};
    std::error_code ec;
    auto canonical = fs::weakly_canonical(path, ec);
    auto &registered = desc.register_file(ec ? path.string() : canonical.string());
    if (has_json_extension(path)) {
        JsonSceneParser{desc, registered, cli_macros}.parse(source);
    } else {
        TextParser{desc, registered, std::move(source), cli_macros}.parse();
    }
}

}// namespace

std::unique_ptr<SceneDesc> parse_scene_file(const std::string &path, const MacroMap &cli_macros) {
    auto desc = std::make_unique<SceneDesc>();
    dispatch_parse(*desc, fs::path{path}, cli_macros);
    return desc;
}

std::unique_ptr<SceneDesc> parse_scene_string(const std::string &source, const std::string &virtual_path,
                                              const MacroMap &cli_macros, bool json) {
    auto desc = std::make_unique<SceneDesc>();
    auto &registered = desc->register_file(virtual_path);
    if (json) {
        JsonSceneParser{*desc, registered, cli_macros}.parse(source);
    } else {
        TextParser{*desc, registered, source, cli_macros}.parse();
    }
    return desc;
}

}// namespace lr
