// stand-in for the fast_float library (src/ext/fast_float is an empty submodule in the reference snapshot).
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim): from_chars for double through strtod -- both are correctly rounded.
#pragma once
#include <system_error>
#include <cstdlib>
#include <string>
namespace fast_float {
struct from_chars_result {
    const char *ptr;
    std::errc ec;
};
inline from_chars_result from_chars(const char *first, const char *last, double &value) noexcept {
    std::string s{first, last};
    char *end = nullptr;
    auto v = std::strtod(s.c_str(), &end);
    if (end == s.c_str()) { return {first, std::errc::invalid_argument}; }
    value = v;
    return {first + (end - s.c_str()), std::errc{}};
}
}// namespace fast_float
