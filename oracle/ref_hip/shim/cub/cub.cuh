#pragma once
#include "../cuda_runtime.h"
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
