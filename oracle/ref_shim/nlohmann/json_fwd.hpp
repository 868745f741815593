// stand-in for nlohmann/json_fwd.hpp (src/ext/json is an empty submodule in the reference snapshot).
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim): libref reads the text scene format only; see ref_json_stub.cpp.
#pragma once
namespace nlohmann {
class json;
}
