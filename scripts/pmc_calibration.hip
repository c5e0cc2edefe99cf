// PMC byte-counter calibration on the render kernels' OWN access patterns (MI355X_MICROARCH.md, HBM section: "calibrate
// on a known byte count in your own access pattern").  Four kernels of KNOWN HBM bytes, each sized past the 256 MiB Infinity Cache, run
// under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: scripts/r6/pmc_calibrate.sh):
//   cal_stream_read    every lane 16 B, coalesced: the guide's calibrated case (FETCH_SIZE reports half of the bytes)
//   cal_stream_write   every lane 16 B, coalesced stores
//   cal_gather48       a wave gathers 64 records of 48 B (three float4 per lane) through a shuffled 4-byte index list, every record ONCE:
//                      what the render kernels' staging does (render_fwd.hip: fwd_walk; render_bwd.hip) without their L2 reuse
//   cal_atomic48       ten lanes add ten floats into a 48-byte record (one global_atomic_add_f32 instruction per instance), shuffled
//                      records, every record once: render_bwd.hip's gradient-record update
// Each prints its known byte counts; the script divides them by the counters.  Test infrastructure, not part of the library.
// build: hipcc --offload-arch=gfx950 -O3 scripts/pmc_calibration.hip -o wild-gaussians_amd/build/pmc_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(256) cal_stream_read(const float4* __restrict__ in, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;   // (never true: keeps the loads)
}
__global__ void __launch_bounds__(256) cal_stream_write(float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void __launch_bounds__(64) cal_gather48(const uint32_t* __restrict__ list, const float4* __restrict__ rec, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (size_t)gridDim.x * 64) {
        const size_t r = 3 * (size_t)list[i];
        const float4 a = rec[r], b = rec[r + 1], c = rec[r + 2];
        acc += a.x + a.w + b.y + b.z + c.x + c.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void __launch_bounds__(64) cal_atomic48(const uint32_t* __restrict__ list, float* __restrict__ rec, size_t n) {
    const int lane = threadIdx.x;
    for (size_t i = (size_t)blockIdx.x; i < n; i += gridDim.x) {   // one instance per wave and step, as the backward walk issues them
        const uint32_t id = list[i];                                // (wave-uniform load: 4 B per instance)
        if (lane < 10) unsafeAtomicAdd(rec + 12 * (size_t)id + lane, 1.0f);
    }
}

int main(int argc, char** argv) {
    const size_t NREC = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (size_t)12 << 20;   // 12 Mi records x 48 B = 576 MiB
    const size_t NSTREAM = (size_t)48 << 20;                                                   // 48 Mi float4 = 768 MiB
    float4 *stream, *rec;
    uint32_t* list;
    float* sink;
    CK(hipMalloc(&stream, NSTREAM * sizeof(float4)));
    CK(hipMalloc(&rec, NREC * 3 * sizeof(float4)));
    CK(hipMalloc(&list, NREC * sizeof(uint32_t)));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(stream, 0, NSTREAM * sizeof(float4)));
    CK(hipMemset(rec, 0, NREC * 3 * sizeof(float4)));
    std::vector<uint32_t> h(NREC);
    for (size_t i = 0; i < NREC; i++) h[i] = (uint32_t)i;
    uint64_t s = 0x9e3779b97f4a7c15ull;   // Fisher-Yates with xorshift64*: a permutation (every record exactly once)
    for (size_t i = NREC - 1; i > 0; i--) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        const size_t j = (size_t)((s * 0x2545f4914f6cdd1dull) % (i + 1));
        std::swap(h[i], h[j]);
    }
    CK(hipMemcpy(list, h.data(), NREC * sizeof(uint32_t), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(cal_stream_read, dim3(4096), dim3(256), 0, 0, stream, NSTREAM, sink);
        hipLaunchKernelGGL(cal_stream_write, dim3(4096), dim3(256), 0, 0, stream, NSTREAM);
        hipLaunchKernelGGL(cal_gather48, dim3(16384), dim3(64), 0, 0, list, rec, NREC, sink);
        hipLaunchKernelGGL(cal_atomic48, dim3(16384), dim3(64), 0, 0, list, (float*)rec, NREC);
        CK(hipDeviceSynchronize());
    }
    std::printf("{\"cal_stream_read\": {\"read_bytes\": %zu, \"write_bytes\": 0}, \"cal_stream_write\": {\"read_bytes\": 0, \"write_bytes\": %zu}, "
                "\"cal_gather48\": {\"read_bytes\": %zu, \"write_bytes\": 0, \"records\": %zu}, "
                "\"cal_atomic48\": {\"read_bytes\": %zu, \"write_bytes\": %zu, \"records\": %zu, \"note\": \"an atomic add reads and writes its 4 bytes at the memory side: 40 B touched per record, counted once each way; + the 4-byte index\"}}\n",
                NSTREAM * sizeof(float4), NSTREAM * sizeof(float4), NREC * 52, NREC, NREC * 44, NREC * 40, NREC);
    return 0;
}
