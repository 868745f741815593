// stand-in for luisa/dsl/syntax.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY): the scalar DSL lives in lc_dsl.h / lc_runtime.h
#pragma once
#include "../lc_runtime.h"
