// Tile binning for gfx950: prefix sum of tiles_touched, (tile|depth) key emission, rocPRIM radix sort,
// per-tile range extraction.  Replaces cub::DeviceScan::InclusiveSum, duplicateWithKeys,
// cub::DeviceRadixSort::SortPairs and identifyTileRanges (rasterizer_impl.cu:70-138, 280, 306-321)
// and the scratch carving of rasterizer_impl.cu:155-194.  Integer work: results are bit-exact.
#include <cstring>
#include <cstdlib>
#include "wg_common.h"

#include <rocprim/rocprim.hpp>

namespace wg {

uint32_t higher_msb(uint32_t n) {  // number of bits needed for values < n, as rasterizer_impl.cu:35-50 computes it
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

size_t query_scan_temp_bytes(size_t P) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, P ? P : 1, rocprim::plus<uint32_t>());
    return bytes;
}

size_t query_sort_temp_bytes(size_t R) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    R ? R : 1, 0, 64);
    return bytes;
}

GeometryState GeometryState::fromChunk(char*& chunk, size_t P) {
    GeometryState g;
    const size_t Pa = P ? P : 1;
    carve(chunk, g.depths, Pa);
    carve(chunk, g.radii, Pa);
    carve(chunk, g.splats, Pa * 3);
    carve(chunk, g.cov3D, Pa * 6);
    carve(chunk, g.clamped, Pa);
    carve(chunk, g.rects, Pa);
    carve(chunk, g.tiles_touched, Pa);
    carve(chunk, g.point_offsets, Pa);
    g.scan_temp_bytes = query_scan_temp_bytes(Pa);
    carve(chunk, g.scan_temp, g.scan_temp_bytes);
    return g;
}

ImageState ImageState::fromChunk(char*& chunk, size_t N, size_t tiles) {
    ImageState img;
    carve(chunk, img.final_T, N ? N : 1);
    carve(chunk, img.n_contrib, N ? N : 1);
    carve(chunk, img.ranges, tiles ? tiles : 1);
    carve(chunk, img.tile_last, tiles ? tiles : 1);
    return img;
}

BinningState BinningState::fromChunk(char*& chunk, size_t R) {
    BinningState b;
    const size_t Ra = R ? R : 1;
    carve(chunk, b.point_list, Ra);
    carve(chunk, b.point_list_unsorted, Ra);
    carve(chunk, b.keys, Ra);
    carve(chunk, b.keys_unsorted, Ra);
    b.sort_temp_bytes = query_sort_temp_bytes(Ra);
    carve(chunk, b.sort_temp, b.sort_temp_bytes);
    return b;
}

hipError_t run_scan(const GeometryState& g, int P, hipStream_t stream) {
    size_t bytes = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_touched, g.point_offsets, (size_t)P, rocprim::plus<uint32_t>(), stream);
}

// One lane per Gaussian; each visible Gaussian writes its run of (key, id) pairs.  Consecutive lanes own
// consecutive runs of the output (offsets are a prefix sum), so a wave's stores land in one contiguous
// window of the key array even though each lane walks its own rectangle.
__global__ void __launch_bounds__(256) duplicate_keys_kernel(int P, const float* __restrict__ depths, const int* __restrict__ radii,
                                                             const ushort4* __restrict__ rects, const uint32_t* __restrict__ offsets,
                                                             uint64_t* __restrict__ keys, uint32_t* __restrict__ values, int gx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    if (radii[idx] > 0) {
        uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
        const ushort4 r = rects[idx];
        const uint32_t depth_bits = __float_as_uint(depths[idx]);
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) {
                uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                key <<= 32;
                key |= depth_bits;
                keys[off] = key;
                values[off] = (uint32_t)idx;
                off++;
            }
    }
}

hipError_t launch_duplicate_keys(int P, const GeometryState& g, const BinningState& b, int gx, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(duplicate_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, g.depths, g.radii, g.rects,
                       g.point_offsets, b.keys_unsorted, b.point_list_unsorted, gx);
    return hipGetLastError();
}

hipError_t run_sort(const BinningState& b, int R, int end_bit, hipStream_t stream) {
    if (R <= 0) return hipSuccess;
    size_t bytes = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.point_list_unsorted, b.point_list, (size_t)R, 0,
                                     (unsigned)end_bit, stream);
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(int L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    const uint32_t curr = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[curr].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (curr != prev) {
            ranges[prev].y = idx;
            ranges[curr].x = idx;
        }
    }
    if (idx == L - 1) ranges[curr].y = L;
}

hipError_t launch_tile_ranges(int R, const BinningState& b, const ImageState& img, int tiles, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(img.ranges, 0, (size_t)tiles * sizeof(uint2), stream);  // rasterizer_impl.cu:313
    if (e != hipSuccess) return e;
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, R, b.keys, img.ranges);
    return hipGetLastError();
}

}  // namespace wg
