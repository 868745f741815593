// stand-in for luisa/core/basic_types.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY)
#pragma once
#include "../lc_types.h"
