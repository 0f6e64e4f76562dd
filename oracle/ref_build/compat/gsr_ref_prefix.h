// Force-included (-include) in front of every reference translation unit.
// Purpose: let hipcc compile the reference's UNMODIFIED CUDA sources, in place, for gfx950.
// It only maps the six CUDA runtime names the sources use onto their HIP equivalents and
// supplies the mixed-signedness min/max overloads nvcc's headers have.  No algorithmic code.
#pragma once
#include <hip/hip_runtime.h>

#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemset hipMemset
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString

// CUDA's math_functions.hpp declares min/max for (unsigned, int) and (int, unsigned):
// the operands are converted to unsigned (auxiliary.h:49-54 relies on it).
__host__ __device__ inline unsigned int min(unsigned int a, int b) { return a < (unsigned int)b ? a : (unsigned int)b; }
__host__ __device__ inline unsigned int min(int a, unsigned int b) { return (unsigned int)a < b ? (unsigned int)a : b; }
__host__ __device__ inline unsigned int max(unsigned int a, int b) { return a > (unsigned int)b ? a : (unsigned int)b; }
__host__ __device__ inline unsigned int max(int a, unsigned int b) { return (unsigned int)a > b ? (unsigned int)a : b; }

// CUDA's __trap(): abort the kernel (auxiliary.h:159).
__host__ __device__ inline void __trap() { __builtin_trap(); }
