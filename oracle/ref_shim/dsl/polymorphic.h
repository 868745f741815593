// stand-in for luisa/dsl/polymorphic.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY): see lc_runtime.h
#pragma once
#include "../lc_runtime.h"
