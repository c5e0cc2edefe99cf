// Probe: which lane holds which value after butterfly10() (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float pair_x32(float a, float b) {
    // (hipcc 7.2 mis-selects "r[0] + r[1]" of __builtin_amdgcn_permlane32_swap as "r[0] + r[0]", so the swap is spelled
    // in asm; the s_nops are the VALU-write -> permlane-swap and permlane-swap -> VALU-read wait states, which hipcc
    // does not insert around inline asm.)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float pair_x16(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float butterfly10(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                             float v8, float v9, int lane) {
    const float w0 = pair_x32(v0, v1), w1 = pair_x32(v2, v3), w2 = pair_x32(v4, v5), w3 = pair_x32(v6, v7), w4 = pair_x32(v8, v9);
    const float u0 = pair_x16(w0, w1), u1 = pair_x16(w2, w3), u2 = pair_x16(w4, w4);
    const bool b0 = lane & 1;
    const float keep0 = b0 ? u1 : u0, send0 = b0 ? u0 : u1;
    const float x0 = keep0 + dpp_f<0xB1>(send0);
    const float x1 = u2 + dpp_f<0xB1>(u2);
    const bool b1 = lane & 2;
    const float keep1 = b1 ? x1 : x0, send1 = b1 ? x0 : x1;
    float y = keep1 + dpp_f<0x4E>(send1);
    y += dpp_f<0x124>(y);
    y += dpp_f<0x128>(y);
    return y;
}
__global__ void probe2(float* out) {
    int lane = threadIdx.x;
    float v[10];
    for (int k = 0; k < 10; k++) v[k] = (float)(k + 1) + 0.001f * lane;  // total = 64*(k+1) + 2.016
    out[lane] = butterfly10(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], lane);
}
__global__ void probe(float* out) {
    int lane = threadIdx.x;
    float v[10];
    for (int k = 0; k < 10; k++) v[k] = (float)((k + 1) * 1000);  // total over 64 lanes = 64000*(k+1)
    out[lane] = pair_x32(v[0], v[1]);
    out[64 + lane] = pair_x16(v[0], v[1]);
    // raw swap semantics
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
    out[128 + lane] = (float)r[0];
    out[192 + lane] = (float)r[1];
    auto q = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
    out[256 + lane] = (float)q[0];
    out[320 + lane] = (float)q[1];
}
int main() {
    float* d; hipMalloc(&d, 384 * 4);
    probe<<<1, 64>>>(d);
    float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"pair_x32(1000,2000)", "pair_x16(1000,2000)", "swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int a = 0; a < 6; a++) { printf("%s:", names[a]); for (int l = 0; l < 64; l++) printf(" %g", h[a * 64 + l]); printf("\n"); }
    probe2<<<1, 64>>>(d);
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    printf("butterfly value index per lane (expected 4*b0+2*b4+b5 | 8+b5):");
    for (int l = 0; l < 64; l++) printf(" %d:%.4f", l, (h[l] - 2.016f) / 64.0f - 1.0f);
    printf("\n");
    return 0;
}
