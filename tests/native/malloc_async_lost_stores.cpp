// Stand-alone reproducer attempt for the runtime defect behind the deterministic backward's leased-block scratch (EXPERIMENTS.md R5.10 /
// R5.12): on ROCm 7.2 / MI355X, kernels writing memory that came from the stream-ordered allocator (hipMallocAsync /
// hipFreeAsync on the device's default pool, release threshold 0: the block is unmapped at every synchronisation and mapped again by the
// next allocation) were seen to LOSE STORES -- inside the library, in the second of three deterministic backward passes over one frame, and,
// with a keep-everything pool, as soon as other host threads called the runtime at the same time.  No product code here: one kernel stores
// a pattern through scattered, partially filled 40-byte slots (what render_bwd.hip's deterministic mode does), a second one verifies it.
//   hipcc --offload-arch=gfx950 -O2 -pthread tests/native/malloc_async_lost_stores.cpp -o wild-gaussians_amd/build/malloc_async_lost_stores
//   malloc_async_lost_stores [iterations=40] [MiB=320] [noise_threads=0|N] [mode: async (default) | malloc]
// Exit code 0 = no store lost in any iteration, 1 = stores were lost (the per-iteration counts are printed), 2 = a HIP call failed.
// `malloc` mode runs the identical kernels on one hipMalloc block (the control: never seen to fail).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9e3779b1u ^ (b + 0x7f4a7c15u);
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    return h;
}
// one wave per slot group: of every 4 slots one is written (the deterministic mode writes the ~quarter of its slots that a tile's pixels
// reached), by ten lanes storing one float each + one flag byte -- the access shape of render_backward_kernel<..., DET>
__global__ void __launch_bounds__(64) store_slots(float* slots, unsigned char* flags, size_t nslots, uint32_t it) {
    const int lane = threadIdx.x;
    for (size_t s = (size_t)blockIdx.x; s < nslots; s += gridDim.x) {
        const uint32_t h = mix((uint32_t)s, it);
        if ((h & 3u) != 0u) continue;          // wave-uniform
        if (lane < 10) slots[s * 10 + lane] = __uint_as_float((h ^ (uint32_t)lane) & 0x7f7fffffu);
        if (lane == 0) flags[s] = 1;
    }
}
__global__ void __launch_bounds__(256) verify_slots(const float* slots, const unsigned char* flags, size_t nslots, uint32_t it, unsigned long long* bad) {
    unsigned long long mine = 0;
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = mix((uint32_t)s, it);
        const bool written = (h & 3u) == 0u;
        if (written != (flags[s] != 0)) { mine++; continue; }
        if (!written) continue;
        for (int k = 0; k < 10; k++)
            if (__float_as_uint(slots[s * 10 + k]) != ((h ^ (uint32_t)k) & 0x7f7fffffu)) { mine++; break; }
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 40;
    const size_t mib = argc > 2 ? (size_t)std::atoll(argv[2]) : 320;
    const int noise = argc > 3 ? std::atoi(argv[3]) : 0;
    const bool use_async = !(argc > 4 && std::strcmp(argv[4], "malloc") == 0);
    const size_t nslots = mib * 1024 * 1024 / 41;   // 40 bytes of sums + 1 flag byte per slot
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned long long* bad;
    CK(hipMalloc(&bad, sizeof(*bad)));
    std::atomic<bool> stop{false};
    std::vector<std::thread> others;
    for (int t = 0; t < noise; t++)   // what the other callers of R5.12 did meanwhile: streams created and destroyed, allocations, small kernels
        others.emplace_back([&stop, t] {
            while (!stop.load()) {
                hipStream_t q;
                if (hipStreamCreate(&q) != hipSuccess) return;
                void* p = nullptr;
                if (hipMallocAsync(&p, (size_t)(8 + 8 * t) << 20, q) == hipSuccess) {
                    (void)hipMemsetAsync(p, t, (size_t)(8 + 8 * t) << 20, q);
                    (void)hipFreeAsync(p, q);
                }
                (void)hipStreamSynchronize(q);
                (void)hipStreamDestroy(q);
            }
        });
    void* fixed = nullptr;
    if (!use_async) CK(hipMalloc(&fixed, nslots * 41));
    int failed_iters = 0;
    for (int it = 0; it < iters; it++) {
        void* p = fixed;
        if (use_async) CK(hipMallocAsync(&p, nslots * 41, s));
        float* slots = static_cast<float*>(p);
        unsigned char* flags = static_cast<unsigned char*>(p) + nslots * 40;
        CK(hipMemsetAsync(flags, 0, nslots, s));          // the flag bytes are cleared, the slot array itself never is (as in the library)
        CK(hipMemsetAsync(bad, 0, sizeof(*bad), s));
        hipLaunchKernelGGL(store_slots, dim3(8160), dim3(64), 0, s, slots, flags, nslots, (uint32_t)it);
        hipLaunchKernelGGL(verify_slots, dim3(2048), dim3(256), 0, s, slots, flags, nslots, (uint32_t)it, bad);
        unsigned long long h = 0;
        CK(hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, s));
        if (use_async) CK(hipFreeAsync(p, s));
        CK(hipStreamSynchronize(s));                      // (default pool, release threshold 0: the block goes back to the system here)
        if (h != 0) { failed_iters++; std::printf("iteration %d: %llu of %zu slots wrong\n", it, h, nslots); }
    }
    stop.store(true);
    for (auto& t : others) t.join();
    std::printf("{\"mode\": \"%s\", \"iterations\": %d, \"MiB\": %zu, \"noise_threads\": %d, \"iterations_with_lost_stores\": %d}\n",
                use_async ? "hipMallocAsync" : "hipMalloc", iters, mib, noise, failed_iters);
    return failed_iters ? 1 : 0;
}
