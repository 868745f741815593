// stand-in for luisa/runtime/rtx/accel.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY): see lc_runtime.h
#pragma once
#include "../../lc_runtime.h"
