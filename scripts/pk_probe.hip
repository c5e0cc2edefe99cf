// Probe: is v_pk_fma_f32 issued at the same rate as v_fma_f32 on gfx950?  (scripts/, not part of the library)
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/pk_probe.hip -o gpurun_out/pk_probe ; run on the GPU box
// (-fno-slp-vectorize is essential: at plain -O3 the SLP vectoriser packs k_scalar's sixteen independent FMAs into eight v_pk_fma_f32 and
//  the probe compares packed with packed -- check with llvm-objdump that k_scalar holds v_fma_f32.  The first repetition runs on cold
//  clocks; read the later ones.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_scalar(float* out, float a, float b, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_packed(float* out, float a, float b, int iters) {
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = v2f{threadIdx.x * 0.001f + 2 * i, threadIdx.x * 0.001f + 2 * i + 1};
    const v2f a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = __builtin_elementwise_fma(x[i], a2, b2);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 2048 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    for (int rep = 0; rep < 5; rep++) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_scalar, dim3(2048), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 16 * iters * 256.0 * 2048;
        printf("scalar fma: %.3f ms  %.1f TFLOP/s\n", ms, flops / ms * 1e-9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_packed, dim3(2048), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("packed fma: %.3f ms  %.1f TFLOP/s\n", ms, flops / ms * 1e-9);
    }
    return 0;
}
