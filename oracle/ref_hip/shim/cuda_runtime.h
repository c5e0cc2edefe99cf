// oracle/ref_hip shim: lets hipcc read the reference's CUDA sources IN PLACE (oracle/ref_hip/Makefile) by giving the CUDA
// header names it includes a HIP body and aliasing the nine CUDA runtime names it uses.  Test infrastructure only: the
// product (wild-gaussians_amd/csrc) never includes this directory.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>  // simple_knn.cu uses FLT_MAX without including it (nvcc's headers happen to)
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMemset hipMemset
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaError_t hipError_t
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define __trap() __builtin_trap()
