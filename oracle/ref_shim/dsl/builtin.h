#pragma once
#include "../lc_runtime.h"
