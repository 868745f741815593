// stand-in for luisa/core/constants.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY): everything lives in lc_core.h
#pragma once
#include "../lc_core.h"
