#pragma once
// glm's platform detection wants a CUDA_VERSION when GLM_FORCE_CUDA is set (the reference sets it,
// forward.h:18); under hipcc glm itself selects GLM_COMPILER_HIP (glm/simd/platform.h:144-145).
#ifndef CUDA_VERSION
#define CUDA_VERSION 11080
#endif
#include <hip/hip_runtime.h>
