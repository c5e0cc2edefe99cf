#pragma once
#include "cuda_runtime.h"
#include <hip/hip_cooperative_groups.h>
