// What does v_min_f32 / fminf / min(float, float) give for a NaN operand on gfx950, as the reference's renderCUDA computes its alpha
// (forward.cu:364: min(0.99f, con_o.w * exp(power)))?  hipcc --offload-arch=gfx950 nan_min_probe.hip -o nan_min_probe && ./nan_min_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, float* out) {
    const float x = in[0], w = in[1];
    out[0] = fminf(0.99f, x);
    out[1] = min(0.99f, x);
    out[2] = min(0.99f, w * exp(x));
    float r;
    asm volatile("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(r) : "v"(x));
    out[3] = r;
    out[4] = exp(x);
    out[5] = (x > 0.0f) ? 1.f : 0.f;
    out[6] = (out[2] < 1.0f / 255.0f) ? 1.f : 0.f;
    out[7] = (float)(int)x;
}
int main() {
    float h[2] = {NAN, 0.0f}, *d, *o, r[8];
    hipMalloc(&d, 8); hipMalloc(&o, 32);
    hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    k<<<1, 1>>>(d, o);
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    printf("fminf(.99,NaN)=%g min(.99f,NaN)=%g min(.99f,0*exp(NaN))=%g v_min_f32=%g exp(NaN)=%g (NaN>0)=%g (alpha<1/255)=%g (int)NaN=%g\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    return 0;
}
