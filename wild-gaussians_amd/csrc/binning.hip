// Tile binning for gfx950.  Replaces cub::DeviceScan::InclusiveSum, duplicateWithKeys, cub::DeviceRadixSort::SortPairs and
// identifyTileRanges (rasterizer_impl.cu:70-138, 280, 306-321) and the scratch carving of rasterizer_impl.cu:155-194.
// Default path: an LDS-only counting sort of the instances by tile (per-chunk tile histograms, column scan, tile scan, scatter of
// 4-byte Gaussian ids into tile buckets) followed by a per-tile register-resident bitonic sort on (depth, id) -- in full, or of a
// depth-nearest front only when lists are long (lazy sort, extended on demand by render_fwd.hip's fix-up kernel).  The
// reference's own scheme -- per-Gaussian prefix sum, (tile|depth) key emission, rocPRIM 64-bit radix sort, range extraction --
// is kept as the fallback for frames whose tile histogram does not fit LDS (> 36 864 tiles) and for tests.
// Integer work: point_list / ranges / num_rendered are bit-exact on every path.
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include "wg_common.h"
#include "wg_sort.h"

#include <rocprim/rocprim.hpp>

namespace wg {

// Number of low key bits the global radix sort has to cover for tile ids < n: the reference's getHigherMsb(n)
// (rasterizer_impl.cu:35-50, a 5-step bisection) equals the position of n's highest set bit, i.e. floor(log2 n) + 1, for every
// n >= 1, and 1 for n == 0 (tests/test_oracle.py holds the literal restatement of the bisection to exactly this formula).
uint32_t higher_msb(uint32_t n) { return n <= 1u ? 1u : 32u - (uint32_t)__builtin_clz(n); }

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

// chunk-local indices are 16-bit, and the near / far split packs (total << 16 | near) per (chunk, tile) histogram word: a chunk must
// hold FEWER than 65536 Gaussians (65536 of them on one tile would carry the near half into the total half), i.e. P <= 65535 * BIN_CHUNKS
bool GeometryState::band_lists_possible(size_t P) { return (P + BIN_CHUNKS - 1) / BIN_CHUNKS <= 65535; }

GeometryState GeometryState::fromChunk(char*& chunk, size_t P, bool lists) {
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
    carve(chunk, g.grad_rec, (Pa * (GRAD_REC_FLOATS + 1) + 3) & ~(size_t)3);   // (+ grad_aux[P] behind the records: the two-colour walk's thirteenth
                                                                               //  sum; a whole number of float4: the clear below writes float4)
    g.scan_temp_bytes = query_scan_temp_bytes(Pa);
    carve(chunk, g.scan_temp, g.scan_temp_bytes);
    lists = lists && band_lists_possible(P);
    carve(chunk, g.band_list, lists ? (size_t)BIN_CHUNKS * 8 * ((Pa + BIN_CHUNKS - 1) / BIN_CHUNKS) : 0);
    carve(chunk, g.band_cnt, lists ? (size_t)BIN_CHUNKS * 16 : 0);   // per (chunk, band): the near (or, without a split, all) candidates, then the far ones
    if (!lists) g.band_list = nullptr;
    return g;
}

ImageState ImageState::fromChunk(char*& chunk, size_t N, size_t tiles) {
    ImageState img;
    carve(chunk, img.final_T, N ? N : 1);
    carve(chunk, img.accum, N ? N : 1);
    carve(chunk, img.n_contrib, N ? N : 1);
    carve(chunk, img.ranges, tiles ? tiles : 1);
    carve(chunk, img.tile_last, tiles ? tiles : 1);
    carve(chunk, img.tile_count, tiles ? tiles : 1);
    carve(chunk, img.tile_offset, tiles + 1);
    carve(chunk, img.chunk_hist, (tiles ? tiles : 1) * (size_t)BIN_CHUNKS);
    carve(chunk, img.order_bwd, tiles ? tiles : 1);
    carve(chunk, img.order_fwd, tiles ? tiles : 1);
    carve(chunk, img.order_key, 4);
    carve(chunk, img.seg_end, tiles ? tiles : 1);
    carve(chunk, img.tile_state, tiles ? tiles : 1);
    carve(chunk, img.stats, 1);
    carve(chunk, img.tile_near, tiles ? tiles : 1);
    carve(chunk, img.far_cursor, tiles ? tiles : 1);
    carve(chunk, img.code_hist, SPLIT_BINS);
    carve(chunk, img.split, 1);
    carve(chunk, img.scan_ticket, 1);
    return img;
}

BinningState BinningState::fromChunk(char*& chunk, size_t R, bool global_sort) {
    BinningState b{};
    const size_t Ra = R ? R : 1;
    carve(chunk, b.point_list, Ra);
    if (!global_sort) {
        carve(chunk, b.bucket_ids, Ra);
    } else {
        carve(chunk, b.point_list_unsorted, Ra);
        carve(chunk, b.keys, Ra);
        carve(chunk, b.keys_unsorted, Ra);
        b.sort_temp_bytes = query_sort_temp_bytes(Ra);
        carve(chunk, b.sort_temp, b.sort_temp_bytes);
    }
    return b;
}

hipError_t run_scan(const GeometryState& g, int P, hipStream_t stream) {
    size_t bytes = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_touched, g.point_offsets, (size_t)P, rocprim::plus<uint32_t>(), stream);
}

// An inclusive scan of non-negative counts that wrapped around 2^32 is not monotone any more: flag it (huge-frame path only).
__global__ void __launch_bounds__(256) scan_overflow_kernel(int P, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t prev = i ? offsets[i - 1] : 0u;
    if (offsets[i] < prev || offsets[i] - prev != counts[i]) *flag = 1u;
}
hipError_t launch_scan_overflow_check(const GeometryState& g, int P, uint32_t* flag, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_overflow_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, g.point_offsets, g.tiles_touched, flag);
    return hipGetLastError();
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

// identifyTileRanges (rasterizer_impl.cu:116-138): only needed when the frame has too many tiles for the LDS histogram.
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

// =====================================================================================================
// Tile-sort path (default).  The (tile | depth) ordering is produced by a counting sort on the tile id
// followed by an independent sort of each tile's bucket in registers.  Global atomics on ~10^4 hot counters are
// memory-side operations on MI355X (measured: 7.4M of them cost 0.4-0.6 ms), so the counting sort only ever
// touches LDS atomics: the Gaussians are cut into BIN_CHUNKS chunks, one workgroup per chunk, and
//   tile_count_kernel          : chunk_hist[c][t] = #instances of chunk c in tile t        (histogram in LDS)
//   chunk_scan_kernel          : chunk_hist[c][t] <- sum_{c' < c} chunk_hist[c'][t];  tile_count[t] = column total
//   tile_scan_kernel           : tile_offset = exclusive_scan(tile_count); ranges; {num_rendered, max_tile_count}
//   tile_scatter_kernel        : cursor[t] (LDS) = tile_offset[t] + chunk_hist[c][t];  bucket_ids[cursor[t]++] = gaussian_id
//   tile_scatter_staged_kernel : the same buckets, laid out tile-major in LDS and written in runs (dense frames)
//   tile_sort_kernel           : one workgroup per tile sorts its bucket (wg_sort.h: bitonic network on 64-bit keys held
//                                in registers) and writes the low words (Gaussian ids) to point_list
//   tile_front_sort_kernel     : when lists are long, only a depth-nearest front of each is extracted and sorted
//                                (lazy sort; the rest is sorted on demand by render_fwd.hip: render_fixup_kernel)
// Sorting (depth_bits, id) ascending inside a tile is exactly the order a stable sort of (tile|depth) keys
// leaves (ties keep ascending Gaussian id, the emission order of duplicateWithKeys), so point_list and
// ranges are bit-identical to the reference's -- whatever order the LDS atomics happened in.  Traffic per
// instance: 4 B scatter + 4 B read + 4 B depth gather (L2) + 4 B write, against 6 radix passes x 24 B for the 45-bit
// global sort.

__device__ __forceinline__ void chunk_bounds(int P, int chunk, int& begin, int& end) {
    const int per = (P + BIN_CHUNKS - 1) / BIN_CHUNKS;
    begin = min(P, chunk * per);
    end = min(P, begin + per);
}

constexpr int BIN_THREADS = 1024;  // count / scatter are chains of dependent LDS atomics: latency-bound, so run 16 waves per chunk

// Visit every tile (x, y) of every lane's rectangle [x0,x1) x [y0,y1): f(x, y, gaussian index).  Convergent: all 64 lanes
// of the wave must call it.  A lane walks a small rectangle itself; a rectangle of more than SERIAL_MAX tiles is handed to
// the whole wave (its corners broadcast with readlane, 8x8 tiles per step).  Without this the kernels' duration was set by
// the largest splat of the frame: one lane issuing ~300 dependent LDS atomics while 63 lanes and the other waves of the
// chip had long finished (rocprofv3: 1.1 resident waves per SIMD on average over the count kernel's 20 us).
template <int SERIAL_MAX, typename F>
__device__ __forceinline__ void for_each_tile(bool valid, int x0, int y0, int x1, int y1, int idx, F f) {
    const int lane = threadIdx.x & 63;
    const int w = valid ? x1 - x0 : 0, h = valid ? y1 - y0 : 0;
    const int n = w * h;
    const bool big = n > SERIAL_MAX;
    if (__ballot(!big && n > 0) != 0ull) {
        // fixed, predicated steps instead of a per-lane loop: the LDS atomics of the steps do not wait for one another
        int x = x0, y = y0;
#pragma unroll
        for (int s = 0; s < SERIAL_MAX; s++) {
            if (!big && s < n) f(x, y, idx);
            x++;
            if (x == x0 + w) { x = x0; y++; }
        }
    }
    uint64_t mask = __ballot(big);
    while (mask != 0ull) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int bx0 = __builtin_amdgcn_readlane(x0, j), by0 = __builtin_amdgcn_readlane(y0, j);
        const int bw = __builtin_amdgcn_readlane(w, j), bh = __builtin_amdgcn_readlane(h, j);
        const int bidx = __builtin_amdgcn_readlane(idx, j);
        const int lx = lane & 7, ly = lane >> 3;
        for (int by = 0; by < bh; by += 8)
            for (int bx = 0; bx < bw; bx += 8)
                if (bx + lx < bw && by + ly < bh) f(bx0 + bx + lx, by0 + by + ly, bidx);
    }
}

constexpr int PF = 8;  // rectangles a thread keeps in flight

// XCD band (wg_alpha.h: xcd_tile) that owns tile t
__device__ __forceinline__ int band_of_tile(int t, int tiles) {
    const int q = tiles >> 3, rem = tiles & 7, split = rem * (q + 1);
    return t < split ? t / (q + 1) : rem + (t - split) / max(q, 1);
}

// With band_list != nullptr (large P) the chunk's Gaussians are also listed per XCD band of tiles they touch (chunk-local 16-bit
// indices, wave-aggregated append): the staged scatter's (chunk, band) workgroups then read their own candidates instead of
// scanning the whole chunk once per band and pass (40 scans of 10 M rectangles at 10 M Gaussians / 4K: 2 of its 2.5 ms).
// With an active near / far split (split != nullptr and a real threshold in it) a histogram word counts two things at once: the
// chunk's instances in the tile in its upper half, the near ones among them in its lower half (a chunk holds fewer than 65536
// Gaussians whenever the split is attempted, so neither half can overflow).
//
// BOX (dense frames / large scenes, where a Gaussian covers tens of tiles): instead of one LDS atomic per (Gaussian, tile) instance
// the rectangle is entered into the histogram as a DIFFERENCE grid -- +w at (x0, y0), -w at (x1, y0) and (x0, y1), +w at (x1, y1),
// corners beyond the grid dropped -- and two prefix passes over the grid (along x, then along y) turn it into the per-tile counts:
// four atomics per Gaussian whatever its size (10 M Gaussians at 4K: 211 M instances -> 40 M atomics).  Exact integer arithmetic
// modulo 2^32, so the packed (total << 16 | near) words come out the same as well: identical histograms.
template <bool BOX, bool LISTS>   // LISTS: band_list != nullptr (a template parameter: with the list code behind a run-time branch the list-free frames' count took 4 us longer)
__global__ void __launch_bounds__(BIN_THREADS) tile_count_kernel(int P, const ushort4* __restrict__ rects, uint32_t* __restrict__ chunk_hist,
                                                         uint16_t* __restrict__ band_list, uint32_t* __restrict__ band_cnt, int gx,
                                                         int tiles, const float* __restrict__ depths, const SplitState* __restrict__ split,
                                                         uint32_t* __restrict__ scan_ticket) {
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t bcnt[16];   // LISTS: per band, the near (without a split: all) candidates listed so far, then the far ones
    const int tid = threadIdx.x, chunk = blockIdx.x, lane = tid & 63;
    if (chunk == 0 && tid == 0) *scan_ticket = 0u;  // the fused column + tile scan's ticket counter (a fresh buffer holds anything)
    const uint32_t near_code = split ? split->near_code : SPLIT_OFF;
    const bool packed = near_code != SPLIT_OFF;  // workgroup-uniform
    for (int t = tid; t < tiles; t += BIN_THREADS) hist[t] = 0;
    if (tid < 16) bcnt[tid] = 0;
    __syncthreads();
    int begin, end;
    chunk_bounds(P, chunk, begin, end);
    const int per = (P + BIN_CHUNKS - 1) / BIN_CHUNKS;
    uint16_t* lists = LISTS ? band_list + (size_t)chunk * 8 * per : nullptr;
    // uniform trip counts (for_each_tile is convergent); PF rectangles are requested before the first is used, so a thread
    // pays one memory latency per PF Gaussians instead of one each (culled Gaussians carry an all-zero rectangle)
    for (int base = begin; base < end; base += PF * BIN_THREADS) {
        ushort4 r[PF];
        uint32_t inc[PF];  // what one instance adds to a histogram word
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int idx = base + k * BIN_THREADS + tid;
            r[k] = idx < end ? rects[idx] : make_ushort4(0, 0, 0, 0);
            inc[k] = 1u;
            if (packed) inc[k] = (idx < end && depth_code(__float_as_uint(depths[idx]), SPLIT_BITS) <= near_code) ? 0x10001u : 0x10000u;
        }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            if (base + k * BIN_THREADS >= end) break;
            const bool valid = r[k].z > r[k].x && r[k].w > r[k].y;
            if constexpr (LISTS) {
                const int b_lo = valid ? band_of_tile(r[k].y * gx + r[k].x, tiles) : 8;
                const int b_hi = valid ? band_of_tile((r[k].w - 1) * gx + r[k].z - 1, tiles) : -1;
                const uint32_t local = (uint32_t)(base + k * BIN_THREADS + tid - begin);
                // the eight bands' appends of this slot in ONE LDS round trip: lane b carries band b's count into a single ds_add_rtn (eight
                // leader-lane atomics one after the other were eight dependent round trips per Gaussian slot, the kernel's chain at 10 M Gaussians:
                // scan 0.231 -> 0.212 ms there, 0.094 -> 0.090 at 3 M.  All 64 (slot, band) pairs of a batch in one atomic: 0.237 / 0.101 -- the sixteen
                // band bounds held across the batch and the ballots taken twice cost more than the seven round trips saved.)
                // With an active split a band's list is kept in two parts: the NEAR Gaussians from the front (what the near scatter reads: about a
                // tenth of the candidates at 10 M Gaussians / 4K -- every candidate costs it a dependent rectangle + depth load), the far ones from
                // the back (what the far scatter reads when a tile of the band asks).  A band has room for the whole chunk: the parts cannot meet.
                const bool nr = (inc[k] & 1u) != 0u;   // (without a split every instance counts 1: one part)
                uint64_t m[8], mf[8];
                uint32_t mine = 0u;
                const uint64_t near_m = __ballot(nr);   // (one ballot for near / far, the parts' masks by scalar AND: not sixteen ballots)
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const uint64_t all = __ballot(b >= b_lo && b <= b_hi);
                    m[b] = all & near_m;
                    mf[b] = all & ~near_m;
                    if (lane == b) mine = (uint32_t)__builtin_popcountll(m[b]);
                    if (lane == 8 + b) mine = (uint32_t)__builtin_popcountll(mf[b]);
                }
                uint32_t wbase = 0u;
                if (lane < 16 && mine != 0u) wbase = atomicAdd(&bcnt[lane], mine);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const bool in = b >= b_lo && b <= b_hi;
                    if (m[b] != 0ull) {   // wave-uniform
                        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)wbase, b);
                        if (in && nr)
                            lists[(size_t)b * per + bb + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[b], 0u))] = (uint16_t)local;
                    }
                    if (mf[b] != 0ull) {  // wave-uniform
                        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)wbase, 8 + b);
                        if (in && !nr)
                            lists[(size_t)b * per + (uint32_t)(per - 1) - (bb + __builtin_amdgcn_mbcnt_hi((uint32_t)(mf[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf[b], 0u)))] = (uint16_t)local;
                    }
                }
            }
            if (BOX) {
                if (valid) {
                    const int gy = tiles / gx;
                    const uint32_t w = inc[k];
                    const int x0 = r[k].x, y0 = r[k].y, x1 = r[k].z, y1 = r[k].w;
                    atomicAdd(&hist[y0 * gx + x0], w);
                    if (x1 < gx) atomicAdd(&hist[y0 * gx + x1], 0u - w);
                    if (y1 < gy) {
                        atomicAdd(&hist[y1 * gx + x0], 0u - w);
                        if (x1 < gx) atomicAdd(&hist[y1 * gx + x1], w);
                    }
                }
            } else {
                for_each_tile<16>(valid, r[k].x, r[k].y, r[k].z, r[k].w, (int)inc[k], [&](int x, int y, int add) { atomicAdd(&hist[y * gx + x], (uint32_t)add); });
            }
        }
    }
    __syncthreads();
    if (BOX) {
        const int gy = tiles / gx, wave = tid >> 6;
        // inclusive prefix along x: one wave per row, 64 columns per step, the carry handed on wave-uniformly
        for (int row = wave; row < gy; row += BIN_THREADS / 64) {
            uint32_t carry = 0;
            for (int xb = 0; xb < gx; xb += 64) {
                const int x = xb + lane;
                uint32_t v = x < gx ? hist[row * gx + x] : 0u;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)v, d);
                    if (lane >= d) v += up;
                }
                v += carry;
                if (x < gx) hist[row * gx + x] = v;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
            }
        }
        __syncthreads();
        // inclusive prefix along y: a column is cut into SEG segments scanned by different threads, then offset by the totals of
        // the segments above (columns are few -- 240 at 4K -- and a single thread per column would be a 135-step chain)
        const int SEG = max(1, min(8, BIN_THREADS / gx)), rows_per = (gy + SEG - 1) / SEG;
        const int col = tid % gx, seg = tid / gx;
        const bool mine = tid < gx * SEG;
        const int ya = seg * rows_per, yb = min(gy, ya + rows_per);
        uint32_t run = 0;
        if (mine)
            for (int y = ya; y < yb; y++) {
                run += hist[y * gx + col];
                hist[y * gx + col] = run;
            }
        __syncthreads();
        // totals of the segments above this one: their last rows (read before anybody adds offsets)
        uint32_t off = 0;
        if (mine)
            for (int sgi = 0; sgi < seg; sgi++) {
                const int last = min(gy, (sgi + 1) * rows_per) - 1;
                if (last >= sgi * rows_per) off += hist[last * gx + col];
            }
        __syncthreads();
        if (mine && off != 0u)
            for (int y = ya; y < yb; y++) hist[y * gx + col] += off;
        __syncthreads();
    }
    uint32_t* out = chunk_hist + (size_t)chunk * tiles;
    for (int t = tid; t < tiles; t += BIN_THREADS) out[t] = hist[t];
    if (LISTS && tid < 16) band_cnt[chunk * 16 + tid] = bcnt[tid];
}

// Column scan: for each tile, exclusive prefix over the chunks.  Workgroup = 16 waves x 64 tiles; wave w owns a sixteenth of the
// chunks, lane = tile (loads are 256-byte coalesced rows of chunk_hist).  The kernel is a chain of dependent row loads per
// wave, so it is cut into many short chains (32 rows each, 8 loads in flight) rather than few long ones.
constexpr int SCAN_WAVES = 16;
// With an active split the words are (total << 16 | near) per chunk: the prefix written back is that of the NEAR counts (the bucket
// positions of the near scatter), tile_count gets the totals, tile_near the near totals.
__device__ __forceinline__ void chunk_scan_body(uint32_t* __restrict__ chunk_hist, uint32_t* tile_count, int tiles,
                                                uint32_t* __restrict__ tile_near, const SplitState* __restrict__ split,
                                                uint32_t (*part)[64], uint32_t (*part_tot)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    constexpr int Q = BIN_CHUNKS / SCAN_WAVES;
    static_assert(Q % 8 == 0, "row batches of 8");
    const bool packed = split && split->near_code != SPLIT_OFF;  // uniform
    uint32_t v[Q];
    uint32_t sum = 0, tot = 0;
    if (t < tiles) {
#pragma unroll
        for (int c = 0; c < Q; c++) v[c] = chunk_hist[(size_t)(wave * Q + c) * tiles + t];
        if (packed) {
#pragma unroll
            for (int c = 0; c < Q; c++) { tot += v[c] >> 16; v[c] &= 0xffffu; }
        }
#pragma unroll
        for (int c = 0; c < Q; c++) sum += v[c];
    }
    part[wave][lane] = sum;
    if (packed) part_tot[wave][lane] = tot;
    __syncthreads();
    uint32_t run = 0;
    for (int w = 0; w < wave; w++) run += part[w][lane];
    if (t < tiles) {
#pragma unroll
        for (int c = 0; c < Q; c++) {
            chunk_hist[(size_t)(wave * Q + c) * tiles + t] = run;
            run += v[c];
        }
        if (wave == SCAN_WAVES - 1) {
            uint32_t total = run;
            if (packed) {
                total = 0;
                for (int w = 0; w < SCAN_WAVES; w++) total += part_tot[w][lane];
            }
            tile_count[t] = total;
            if (tile_near) tile_near[t] = run;
        }
    }
}

__global__ void __launch_bounds__(64 * SCAN_WAVES) chunk_scan_kernel(uint32_t* __restrict__ chunk_hist, uint32_t* __restrict__ tile_count, int tiles,
                                                                     uint32_t* __restrict__ tile_near, const SplitState* __restrict__ split) {
    __shared__ uint32_t part[SCAN_WAVES][64];
    __shared__ uint32_t part_tot[SCAN_WAVES][64];
    chunk_scan_body(chunk_hist, tile_count, tiles, tile_near, split, part, part_tot);
}

// PER = tiles per thread, a compile-time bound so that a thread's counts are loaded together (the kernel is one workgroup on a
// critical path: its duration is its chain of memory round trips) and kept in registers for the second pass.
template <int PER>
__device__ __forceinline__ void tile_scan_body(const uint32_t* tile_count, uint32_t* __restrict__ tile_offset,
                                               uint2* __restrict__ ranges, BinStats* __restrict__ stats, int tiles,
                                               HostMailbox* mailbox, uint32_t seq, const SplitState* __restrict__ split, SpecLimits spec,
                                               uint32_t* wave_sum, uint32_t* wave_max, uint32_t* wave_ovf) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The instance total is a 32-bit sum (the reference's is a 32-bit int, rasterizer_impl.cu:280-284, and overflows silently).
    // Every addition below is checked for wrap-around: if none wraps, every partial sum is exact, so a total of 2^32 or more
    // always trips the flag; the kernel then reports 0xffffffff and the host returns WG_ERR_OVERFLOW before any buffer is sized.
    // Layout: wave w owns the contiguous segment [w * seg, (w + 1) * seg) of the tiles, seg = 64 * per; in step k its lanes take the
    // 64 consecutive tiles seg * w + 64 k + lane -- every load and store is one coalesced line per wave (round 2 gave a THREAD `per`
    // consecutive tiles: 64 lines per wave-load, 53 us for the 32 400 tiles of a 4K frame) -- and a wave-level scan per step carries
    // the running prefix along the segment.
    bool ovf = false;
    const int per = (tiles + 1023) / 1024;  // <= PER steps
    const int seg0 = wave * 64 * per;
    uint32_t cnt[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = seg0 + 64 * k + lane;
        cnt[k] = (k < per && t < tiles) ? tile_count[t] : 0u;
    }
    uint32_t carry = 0, lmax = 0;   // (cnt[k] turns into the exclusive prefix of the lane's tile of step k INSIDE the wave's segment)
#pragma unroll
    for (int k = 0; k < PER; k++) {
        if (k < per) {   // workgroup-uniform
            uint32_t incl = cnt[k];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) {
                    incl += up;
                    ovf |= incl < up;
                }
            }
            lmax = max(lmax, cnt[k]);
            cnt[k] = carry + (incl - cnt[k]);
            ovf |= cnt[k] < carry;
            const uint32_t step_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            carry += step_total;
            ovf |= carry < step_total;
        }
    }
    const uint32_t incl = carry;   // the segment's total (wave-uniform)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, m));
    if (lane == 0) wave_sum[wave] = incl;
    if (lane == 0) wave_max[wave] = lmax;
    const uint64_t any_ovf = __ballot(ovf);
    if (lane == 0) wave_ovf[wave] = any_ovf != 0ull ? 1u : 0u;
    __syncthreads();
    uint32_t wave_base = 0, total = 0, gmax = 0, tovf = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) wave_base += wave_sum[w];
        total += wave_sum[w];
        tovf |= (total < wave_sum[w]) ? 1u : 0u;
        tovf |= wave_ovf[w];
        gmax = max(gmax, wave_max[w]);
    }
    if (tovf) total = 0xffffffffu;
    if (tid == 0) {  // first: the two numbers the host is waiting for (it launches the next kernels behind this one)
        tile_offset[tiles] = total;
        const uint32_t active = (split && split->near_code != SPLIT_OFF) ? 1u : 0u;
        stats->num_rendered = total;
        stats->max_tile_count = gmax;
        stats->split_active = active;
        // speculative forward: does the frame fit what the host has already enqueued behind this kernel?
        const uint32_t fail = (spec.capacity != 0u && (total > spec.capacity || gmax > spec.max_list)) ? 1u : 0u;
        stats->spec_fail = fail;
        if (mailbox) {
            const unsigned long long hi = (unsigned long long)seq << 32;
            __hip_atomic_store(&mailbox->word0, hi | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mailbox->word1, hi | ((unsigned long long)fail << 31) | ((unsigned long long)active << 30) | min(gmax, MAILBOX_MAX_LIST),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = seg0 + 64 * k + lane;
        if (k < per && t < tiles) {
            const uint32_t c = tile_count[t], run = wave_base + cnt[k];   // (the count once more, a coalesced L2 hit: 36 registers less)
            tile_offset[t] = run;
            ranges[t] = c ? make_uint2(run, run + c) : make_uint2(0u, 0u);  // empty tiles stay (0,0) as after the reference's memset
        }
    }
}

// One XCD band's tiles [start, start + cnt) in (roughly) descending cost order: a 256-bin counting sort by one 1024-thread workgroup.
// period == 0: plain descending order (the backward kernel: its second residency round is dealt dynamically, shortest tiles last).
// period > 0 (the forward kernel, whose ~8 k waves are ALL resident at once): the hardware places a band's workgroups on its XCD's 128
// SIMDs with period 128 (workgroups i and i + 128 of the band share a SIMD: measured with the probe build, scripts/probe_balance.py --
// 83 - 100 % of the SIMDs exactly, the rest one phase jump of 32), so the sorted tiles are dealt in rounds of `period`, every other round
// reversed ("snake"): each SIMD gets one tile of every size class and the classes' slopes cancel.
template <typename CostFn>
__device__ __forceinline__ void order_band(CostFn cost_of, uint32_t start, uint32_t cnt, uint32_t* __restrict__ order, uint32_t period,
                                           uint32_t* hist, uint32_t* wmax) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t m = 0;
    for (uint32_t i = tid; i < cnt; i += 1024) m = max(m, cost_of(i));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if (lane == 0) wmax[wave] = m;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    m = 0;
    for (int w = 0; w < 16; w++) m = max(m, wmax[w]);
    const float scale = 255.0f / (float)max(m, 1u);
    auto bin_of = [&](uint32_t c) { return 255u - min(255u, (uint32_t)((float)c * scale)); };  // bin 0 = the most expensive tiles
    for (uint32_t i = tid; i < cnt; i += 1024) atomicAdd(&hist[bin_of(cost_of(i))], 1u);
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the 256 bins: 4 per lane
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = hist[4 * tid + k]; sum += v[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) { hist[4 * tid + k] = run; run += v[k]; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += 1024) {
        uint32_t rank = atomicAdd(&hist[bin_of(cost_of(i))], 1u);
        if (period != 0u) {
            const uint32_t r = rank / period, j = rank - r * period;
            const uint32_t len = min(period, cnt - r * period);   // (the last round of a band is a partial one)
            rank = r * period + ((r & 1u) ? len - 1u - j : j);
        }
        order[start + rank] = start + i;
    }
}

// ---- launch order of the FORWARD render kernel: tile costs remembered per camera ------------------------------------------------------
// The forward kernel's tile costs (the walked lengths, tile_last) are only known once it has run -- but training renders the same cameras
// over and over while the Gaussians move slowly, and an earlier frame of the same camera predicts them well.  The library keeps, per device, a
// direct-mapped table in device memory (api.hip: OrderTable): row = hash(viewmatrix, projmatrix, W, H) -> {64-bit tag, uint32 cost[tiles]}.
// Everything happens ON THE DEVICE (the camera matrices are device memory: the host never sees their values): eight workgroups -- one per
// XCD band, riding along in the launch of the single-workgroup tile scan, which leaves the chip idle anyway -- hash the camera, look the
// row up and write the launch order -- order_band()'s snake on a hit, the identity on a miss -- plus the row's index into the frame's image
// state; the render kernel writes every tile's walked length (and the tag) back into the row.  Pure scheduling: a stale, torn or colliding
// row costs balance, never a result.
__device__ __forceinline__ void forward_order_body(int band, const FwdOrderArgs& a, int tiles, uint32_t* hist, uint32_t* wmax, uint32_t* s_key) {
    if (threadIdx.x == 0) {
        unsigned long long h = 0xcbf29ce484222325ull;   // FNV-1a over the 34 words
        auto mix = [&](uint32_t w) { h = (h ^ (unsigned long long)w) * 0x100000001b3ull; };
        for (int i = 0; i < 16; i++) mix(__float_as_uint(a.viewmatrix[i]));
        for (int i = 0; i < 16; i++) mix(__float_as_uint(a.projmatrix[i]));
        mix((uint32_t)a.W); mix((uint32_t)a.H);
        h ^= h >> 29;
        const uint32_t per = ((uint32_t)tiles + a.stride - 1u) / a.stride;   // table rows a frame of this size takes
        const uint32_t groups = a.slots / per;
        const uint32_t slot = (uint32_t)(h % (unsigned long long)groups) * per;
        const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32) | 1u;   // (hi != 0: an empty row never matches)
        const uint32_t* tag = a.table + 2 * (size_t)slot;
        s_key[0] = slot; s_key[1] = lo; s_key[2] = hi;
        s_key[3] = (tag[0] == lo && tag[1] == hi) ? 1u : 0u;
        if (band == 0) { a.key_out[0] = slot; a.key_out[1] = lo; a.key_out[2] = hi; a.key_out[3] = s_key[3]; }
    }
    __syncthreads();
    const int q = tiles >> 3, rem = tiles & 7, x = band;
    const uint32_t start = x * q + min(x, rem), cnt = q + (x < rem ? 1 : 0);
    if (s_key[3] == 0u) {   // no earlier frame of this camera: image order
        for (uint32_t i = threadIdx.x; i < cnt; i += 1024) a.order[start + i] = start + i;
        return;
    }
    const uint32_t* cost = a.table + 2 * (size_t)a.slots + (size_t)s_key[0] * a.stride;
    order_band([&](uint32_t i) { return cost[start + i]; }, start, cnt, a.order, a.period, hist, wmax);
}

// stand-alone form (frames whose tile scan takes another path: the fused scan, more tiles than the LDS histogram holds)
__global__ void __launch_bounds__(1024) forward_order_kernel(FwdOrderArgs a, int tiles) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wmax[16];
    __shared__ uint32_t s_key[4];
    forward_order_body((int)blockIdx.x, a, tiles, hist, wmax, s_key);
}

hipError_t launch_forward_order(const FwdOrderArgs& a, int tiles, hipStream_t stream) {
    if (tiles <= 0 || !a.table) return hipSuccess;
    hipLaunchKernelGGL(forward_order_kernel, dim3(8), dim3(1024), 0, stream, a, tiles);
    return hipGetLastError();
}

template <int PER>
__global__ void __launch_bounds__(1024) tile_scan_kernel(const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_offset,
                                                         uint2* __restrict__ ranges, BinStats* __restrict__ stats, int tiles,
                                                         HostMailbox* mailbox, uint32_t seq, const SplitState* __restrict__ split, SpecLimits spec,
                                                         FwdOrderArgs fo) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t wave_max[16];
    __shared__ uint32_t wave_ovf[16];
    if (blockIdx.x != 0) {   // workgroups 1..8 (present when fo.table != nullptr): the forward render kernel's launch order, one XCD band each
        __shared__ uint32_t hist[256];
        __shared__ uint32_t s_key[4];
        forward_order_body((int)blockIdx.x - 1, fo, tiles, hist, wave_sum, s_key);
        return;
    }
    tile_scan_body<PER>(tile_count, tile_offset, ranges, stats, tiles, mailbox, seq, split, spec, wave_sum, wave_max, wave_ovf);
}

// chunk_scan + tile_scan in ONE launch (option "fused_scan"): every column-scan workgroup publishes its 64 tile totals with a
// device-scope release and takes a ticket; the workgroup that draws the last ticket acquires and runs the tile scan.  Saves a launch
// on the forward pass's critical chain; costs the device-scope fences (an L2 write-back per workgroup on this multi-XCD part).
template <int PER>
__global__ void __launch_bounds__(64 * SCAN_WAVES) chunk_tile_scan_kernel(uint32_t* __restrict__ chunk_hist, uint32_t* tile_count, int tiles,
                                                                          uint32_t* __restrict__ tile_near, const SplitState* __restrict__ split,
                                                                          uint32_t* __restrict__ ticket, uint32_t* __restrict__ tile_offset,
                                                                          uint2* __restrict__ ranges, BinStats* __restrict__ stats,
                                                                          HostMailbox* mailbox, uint32_t seq, SpecLimits spec) {
    __shared__ uint32_t part[SCAN_WAVES][64];
    __shared__ uint32_t part_tot[SCAN_WAVES][64];
    __shared__ uint32_t s_last;
    chunk_scan_body(chunk_hist, tile_count, tiles, tile_near, split, part, part_tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // release: this workgroup's tile_count entries before its ticket
        s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;
    __threadfence();  // acquire: every other workgroup's entries
    if (threadIdx.x == 0) *ticket = 0u;  // ready for the next frame that carves the same buffer
    uint32_t* wave_sum = &part[0][0];
    uint32_t* wave_max = &part[1][0];
    uint32_t* wave_ovf = &part[2][0];
    tile_scan_body<PER>(tile_count, tile_offset, ranges, stats, tiles, mailbox, seq, split, spec, wave_sum, wave_max, wave_ovf);
}

// Scatter of the Gaussian ids into the tile buckets.  Each workgroup owns one (Gaussian chunk, XCD band of tiles) pair
// and only emits the instances that fall into its band.  4-byte stores to random positions of a 30 MB array leave the
// L2 as partial lines (measured: 238 MB of fabric writes for 30 MB of payload when every workgroup wrote everywhere);
// with the band = the XCD the dispatcher is observed to run the block on (block b -> XCD b % 8), an XCD's L2 only ever
// holds its own eighth of the bucket array (< 4 MiB at 1080p) and lines leave it complete.  The band assignment is a
// speed matter only: any placement gives the same buckets.
// INSTANCE-parallel since round 3.  Rounds 1-2 walked rectangles: a (chunk, band) workgroup visited all of the chunk's Gaussians,
// one per lane and round, sixteen predicated steps each, although only one in five has a row inside the band and a valid one emits
// ~5 instances there -- 5 % of the lane-steps did anything (PMC: 1 500 vector instructions per wave for 450 emitted instances; with
// every store and atomic removed that kernel still took 41 of its 52 us).  Now the workgroup (1) appends the Gaussians that do reach
// its band to a list in LDS (clipped rectangle, entry, instance count), (2) scans the counts, and (3) deals the band's instances
// out evenly: a thread takes a contiguous run of them, finds its first Gaussian with one binary search of the prefix and walks on
// from there.  Every lane-step emits; a splat of 300 tiles is spread over the workgroup like any other run.  0.0575 -> 0.0478 ms
// at the headline scene, bit-identical outputs (profiles/r3/ab_scatter_instance_parallel.txt); what is left is the L2's rate of
// 4-byte store requests (~0.9 M per XCD) and the workgroups' start-up chain.  16 KB of list per workgroup: with 32 KB (eight
// candidates per thread and pass) the kernel loses its occupancy and the whole gain.
// (Measured and dropped earlier: staging a workgroup's instances tile-major in LDS and writing them out in runs brought the fabric
// writes down to the 30 MB of payload and was not faster at ~900 instances per tile -- it is the kernel for dense frames, below.)
constexpr int SC_PF = 4;                 // candidates a thread looks at per pass
constexpr int SC_CAP = SC_PF * 256;      // list capacity of a pass (every candidate of the pass may reach the band)
template <bool LISTS>
__global__ void __launch_bounds__(256) tile_scatter_kernel(int P, const ushort4* __restrict__ rects, const float* __restrict__ depths,
                                                           const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ chunk_hist,
                                                           const uint16_t* __restrict__ band_list, const uint32_t* __restrict__ band_cnt,
                                                           uint32_t* __restrict__ bucket_ids, int gx, int tiles, int code_bits,
                                                           const SplitState* __restrict__ split, const BinStats* __restrict__ guard) {
    extern __shared__ uint32_t cursor[];
    __shared__ uint32_t q_entry[SC_CAP], q_xy[SC_CAP], q_wh[SC_CAP], q_pre[SC_CAP];  // per listed Gaussian: entry, x0 | ya << 16, w | h << 16, exclusive prefix of w * h
    __shared__ uint32_t q_n, wave_tot[4];
    if (guard && guard->spec_fail) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, band = blockIdx.x & 7, chunk = blockIdx.x >> 3;
    const uint32_t near_code = split ? split->near_code : SPLIT_OFF;
    const bool near_only = near_code != SPLIT_OFF;
    const int q = tiles >> 3, rem = tiles & 7;
    const int t0 = band * q + min(band, rem), t1 = t0 + q + (band < rem ? 1 : 0);
    if (t0 >= t1) return;
    int begin, end;
    chunk_bounds(P, chunk, begin, end);
    const int per = (P + BIN_CHUNKS - 1) / BIN_CHUNKS;
    const uint16_t* list = LISTS ? band_list + ((size_t)chunk * 8 + band) * per : nullptr;
    const int cn = LISTS ? (int)band_cnt[chunk * 16 + band] : end - begin;   // (with an active split: the near candidates, the front part of the list)
    if (cn == 0) return;
    const uint32_t* hbase = chunk_hist + (size_t)chunk * tiles;
    for (int t = t0 + tid; t < t1; t += 256) cursor[t - t0] = tile_offset[t] + hbase[t];
    const int y0 = t0 / gx, y1 = (t1 - 1) / gx;
    for (int base = 0; base < cn; base += SC_CAP) {
        if (tid == 0) q_n = 0;
        __syncthreads();  // (the first pass: the cursors; later ones: the previous pass's list has been read)
        // (1) the candidates that reach the band
        ushort4 r[SC_PF];
        uint32_t entry[SC_PF];
#pragma unroll
        for (int k = 0; k < SC_PF; k++) {
            const int i = base + k * 256 + tid;
            const int idx = i < cn ? begin + (LISTS ? (int)list[i] : i) : -1;
            r[k] = idx >= 0 ? rects[idx] : make_ushort4(0, 0, 0, 0);
            entry[k] = idx >= 0 ? (uint32_t)idx : 0u;
            if ((code_bits || near_only) && idx >= 0) {
                const uint32_t db = __float_as_uint(depths[idx]);
                if (code_bits) entry[k] |= depth_code(db, (uint32_t)code_bits) << (32 - code_bits);
                if (near_only && depth_code(db, SPLIT_BITS) > near_code) r[k] = make_ushort4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < SC_PF; k++) {
            if (base + k * 256 >= cn) break;  // workgroup-uniform
            const int ya = max((int)r[k].y, y0), yb = min((int)r[k].w, y1 + 1);
            const bool valid = r[k].z > r[k].x && yb > ya;
            const uint64_t m = __ballot(valid);
            if (m != 0ull) {  // wave-aggregated append
                const int leader = __builtin_ctzll(m);
                uint32_t wbase = 0;
                if (lane == leader) wbase = atomicAdd(&q_n, (uint32_t)__builtin_popcountll(m));
                wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, leader);
                if (valid) {
                    const uint32_t pos = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    q_entry[pos] = entry[k];
                    q_xy[pos] = (uint32_t)r[k].x | ((uint32_t)ya << 16);
                    q_wh[pos] = (uint32_t)(r[k].z - r[k].x) | ((uint32_t)(yb - ya) << 16);
                }
            }
        }
        __syncthreads();
        const uint32_t n = q_n;
        if (n == 0u) continue;  // workgroup-uniform
        // (2) exclusive prefix of the listed Gaussians' instance counts: SC_PF consecutive entries per thread
        uint32_t c[SC_PF], sum = 0;
#pragma unroll
        for (int e = 0; e < SC_PF; e++) {
            const uint32_t j = (uint32_t)SC_PF * (uint32_t)tid + e;
            const uint32_t wh = j < n ? q_wh[j] : 0u;
            c[e] = (wh & 0xffffu) * (wh >> 16);
            sum += c[e];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = incl - sum, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t wt = wave_tot[w];
            if (w < wave) before += wt;
            total += wt;
        }
#pragma unroll
        for (int e = 0; e < SC_PF; e++) {
            const uint32_t j = (uint32_t)SC_PF * (uint32_t)tid + e;
            if (j < n) q_pre[j] = before;
            before += c[e];
        }
        __syncthreads();
        // (3) the band's `total` instances of this pass, a contiguous run per thread
        const uint32_t run = (total + 255u) >> 8;
        uint32_t i = min(total, run * (uint32_t)tid);
        const uint32_t i_end = min(total, i + run);
        if (i < i_end) {
            uint32_t lo = 0, hi = n - 1;  // the last listed Gaussian whose prefix is <= i (counts are >= 1: prefixes strictly increase)
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (q_pre[mid] <= i) lo = mid; else hi = mid - 1u;
            }
            uint32_t j = lo;
            uint32_t xy = q_xy[j], wh = q_wh[j], ent = q_entry[j];
            int x0 = (int)(xy & 0xffffu), w = (int)(wh & 0xffffu);
            const uint32_t o = i - q_pre[j];
            int y = (int)(xy >> 16) + (int)(o / (uint32_t)w), x = x0 + (int)(o % (uint32_t)w);
            int yb = (int)(xy >> 16) + (int)(wh >> 16);
            for (; i < i_end; i++) {
                const int t = y * gx + x;
                if (t >= t0 && t < t1) {
                    const uint32_t pos = atomicAdd(&cursor[t - t0], 1u);
                    bucket_ids[pos] = ent;
                }
                if (++x == x0 + w) {
                    x = x0;
                    if (++y == yb && i + 1u < i_end) {  // on to the next listed Gaussian
                        j++;
                        xy = q_xy[j]; wh = q_wh[j]; ent = q_entry[j];
                        x0 = x = (int)(xy & 0xffffu); w = (int)(wh & 0xffffu);
                        y = (int)(xy >> 16); yb = y + (int)(wh >> 16);
                    }
                }
            }
        }
    }
}

// Staged scatter, for dense frames.  The direct kernel issues one 4-byte store per instance to a random bucket position; with
// thousands of instances per tile that is what bounds it (41.8 M instances: 0.44 ms, 4.5x write amplification at the fabric).
// Here a workgroup owns (G consecutive chunks, XCD band), knows from the chunk histograms how many instances it will emit per
// tile, lays them out tile-major in LDS (LDS cursors), and copies every tile's run to its bucket with consecutive lanes on
// consecutive addresses: one write transaction per (workgroup, tile) run.  At ~900 instances per tile the runs are 2 ids long
// and the direct kernel is as fast (measured), so the host picks this one only for long lists.  Placement inside a bucket is
// free: the same multiset either way.
template <bool LISTS>  // a template parameter, not a runtime test: a branch in the prefetch loop makes the compiler wait at every join
__global__ void __launch_bounds__(1024) tile_scatter_staged_kernel(int P, const ushort4* __restrict__ rects, const float* __restrict__ depths,
                                                                  const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ tile_count,
                                                                  const uint32_t* __restrict__ chunk_hist, const uint16_t* __restrict__ band_list,
                                                                  const uint32_t* __restrict__ band_cnt, uint32_t* __restrict__ bucket_ids,
                                                                  int gx, int tiles, int G, int nbmax, uint32_t cap, int code_bits,
                                                                  const SplitState* __restrict__ split, const BinStats* __restrict__ guard) {
    extern __shared__ uint32_t smem[];
    if (guard && guard->spec_fail) return;  // workgroup-uniform
    // active split: only near Gaussians are emitted; chunk_hist holds the prefix of the near counts and `tile_count` is tile_near
    const uint32_t near_code = split ? split->near_code : SPLIT_OFF;
    const bool near_only = near_code != SPLIT_OFF;
    uint32_t* gbase = smem;           // [nbmax] bucket position of this workgroup's first instance of the tile
    uint32_t* lcur = gbase + nbmax;   // [nbmax] cursor: staging-area positions (staged) or bucket positions (direct)
    uint32_t* wsum = lcur + nbmax;    // [16]
    uint32_t* stage = wsum + 16;      // [cap]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int band = blockIdx.x & 7, c0 = (blockIdx.x >> 3) * G, c1 = c0 + G;
    const int q = tiles >> 3, rem = tiles & 7;
    const int t0 = band * q + min(band, rem), t1 = t0 + q + (band < rem ? 1 : 0);
    const int nb = t1 - t0;
    if (nb <= 0) return;
    for (int i = tid; i < nb; i += 1024) {
        const int t = t0 + i;
        const uint32_t pre0 = chunk_hist[(size_t)c0 * tiles + t];
        const uint32_t pre1 = c1 < BIN_CHUNKS ? chunk_hist[(size_t)c1 * tiles + t] : tile_count[t];
        gbase[i] = tile_offset[t] + pre0;
        lcur[i] = pre1 - pre0;
    }
    __syncthreads();
    // exclusive scan of the per-tile counts -> run positions in the staging area
    const int per = (nb + 1023) / 1024;
    const int sb = min(nb, tid * per), se = min(nb, sb + per);
    uint32_t local = 0;
    for (int i = sb; i < se; i++) local += lcur[i];
    uint32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) wave_base += wsum[w];
        total += wsum[w];
    }
    uint32_t run = wave_base + incl - local;
    for (int i = sb; i < se; i++) {
        const uint32_t c = lcur[i];
        lcur[i] = run;  // start of tile i's run in the workgroup's (virtual) tile-major sequence
        run += c;
    }
    __syncthreads();

    // The sequence is emitted in passes of as many whole tiles as fit the staging area (one pass when the share is small; a
    // dense frame takes several, re-reading the chunk's rectangles each time).  A single tile whose run alone exceeds the area
    // is stored directly.
    for (int a = 0; a < nb;) {
        const uint32_t base = lcur[a];  // untouched so far: cursors of tiles >= a have not been advanced
        int e;
        {  // largest e with start[e] - base <= cap (start[nb] = total); at least one tile
            int lo = a + 1, hi = nb;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((mid < nb ? lcur[mid] : total) - base <= cap) lo = mid;
                else hi = mid - 1;
            }
            e = lo;
        }
        const uint32_t pass_total = (e < nb ? lcur[e] : total) - base;
        const bool staged = pass_total <= cap;  // false only for a single oversized tile
        __syncthreads();                        // everybody has read the cursors this pass' scatter is about to advance
        const int ta = t0 + a, tb = t0 + e;
        const int ya0 = ta / gx, yb0 = (tb - 1) / gx;
        if (pass_total > 0) {
            // candidates: the band's list of every chunk of the group (large P), or the chunks themselves
            const int per = (P + BIN_CHUNKS - 1) / BIN_CHUNKS;
            for (int c = c0; c < c1; c++) {
                int cb, ce;
                chunk_bounds(P, c, cb, ce);
                const uint16_t* list = LISTS ? band_list + ((size_t)c * 8 + band) * per : nullptr;
                const int cn = LISTS ? (int)band_cnt[c * 16 + band] : ce - cb;
                for (int gb0 = 0; gb0 < cn; gb0 += PF * 1024) {
                    ushort4 r[PF];
                    uint32_t entry[PF];
                    int idx[PF];
#pragma unroll
                    for (int k = 0; k < PF; k++) {
                        const int i = gb0 + k * 1024 + tid;
                        idx[k] = i < cn ? cb + (LISTS ? (int)list[i] : i) : -1;
                    }
#pragma unroll
                    for (int k = 0; k < PF; k++) {
                        const bool in = idx[k] >= 0;
                        r[k] = in ? rects[idx[k]] : make_ushort4(0, 0, 0, 0);
                        entry[k] = in ? (uint32_t)idx[k] : 0u;
                        if ((code_bits || near_only) && in) {
                            const uint32_t db = __float_as_uint(depths[idx[k]]);
                            if (code_bits) entry[k] |= depth_code(db, (uint32_t)code_bits) << (32 - code_bits);
                            if (near_only && depth_code(db, SPLIT_BITS) > near_code) r[k] = make_ushort4(0, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < PF; k++) {
                        if (gb0 + k * 1024 >= cn) break;
                        const int ya = max((int)r[k].y, ya0), yb = min((int)r[k].w, yb0 + 1);
                        for_each_tile<16>(r[k].z > r[k].x && yb > ya, r[k].x, ya, r[k].z, yb, (int)entry[k], [&](int x, int y, int id) {
                            const int t = y * gx + x;
                            if (t >= ta && t < tb) {
                                const uint32_t slot = atomicAdd(&lcur[t - t0], 1u) - base;
                                if (staged) stage[slot] = (uint32_t)id;
                                else bucket_ids[gbase[t - t0] + slot] = (uint32_t)id;
                            }
                        });
                    }
                }
            }
        }
        __syncthreads();
        if (staged && pass_total > 0) {
            // copy-out, 16 lanes per run: after the scatter lcur[i] is the END of tile i's run, i.e. the start of tile i+1's
            const int grp = tid >> 4, sub = tid & 15;
            for (int i = a + grp; i < e; i += 64) {
                const uint32_t lb = (i == a ? base : lcur[i - 1]) - base, n = lcur[i] - base - lb, gb = gbase[i];
                for (uint32_t k = sub; k < n; k += 16) bucket_ids[gb + k] = stage[lb + k];
            }
            __syncthreads();  // the staging area is reused by the next pass
        }
        a = e;
    }
}

// ---- near / far split ---------------------------------------------------------------------------------------------------
// Far scatter: after the first fix-up phase, the tiles whose pixels are still accumulating with every near instance consumed
// (tile_state != 0xffffffff) get the instances of the far Gaussians, appended behind the near ones through a per-tile global
// cursor (the slow kind of atomic, but this is the rare path).  A workgroup leaves at once when no tile of its band asked.
template <bool LISTS>
__global__ void __launch_bounds__(256) tile_scatter_far_kernel(int P, const ushort4* __restrict__ rects, const float* __restrict__ depths,
                                                               const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ tile_near,
                                                               const uint32_t* __restrict__ tile_state, uint32_t* __restrict__ far_cursor,
                                                               const uint16_t* __restrict__ band_list, const uint32_t* __restrict__ band_cnt,
                                                               uint32_t* __restrict__ bucket_ids, int gx, int tiles, int code_bits,
                                                               const SplitState* __restrict__ split, const BinStats* __restrict__ guard) {
    const int tid = threadIdx.x, band = blockIdx.x & 7, chunk = blockIdx.x >> 3;
    if (guard && guard->spec_fail) return;
    if (((split->need_far >> band) & 1u) == 0u || split->near_code == SPLIT_OFF) return;  // no tile of this band asked
    const uint32_t near_code = split->near_code;
    const int q = tiles >> 3, rem = tiles & 7;
    const int t0 = band * q + min(band, rem), t1 = t0 + q + (band < rem ? 1 : 0);
    if (t0 >= t1) return;
    int begin, end;
    chunk_bounds(P, chunk, begin, end);
    const int per = (P + BIN_CHUNKS - 1) / BIN_CHUNKS;
    // (the far candidates: the back part of the band's list, tile_count_kernel)
    const int cn = LISTS ? (int)band_cnt[chunk * 16 + 8 + band] : end - begin;
    const uint16_t* list = LISTS ? band_list + ((size_t)chunk * 8 + band) * per + (per - cn) : nullptr;
    const int y0 = t0 / gx, y1 = (t1 - 1) / gx;
    for (int base = 0; base < cn; base += 256) {  // uniform trip count: for_each_tile is convergent
        const int i = base + tid;
        const int idx = i < cn ? begin + (LISTS ? (int)list[i] : i) : -1;
        ushort4 r = make_ushort4(0, 0, 0, 0);
        uint32_t entry = 0u;
        if (idx >= 0) {
            const uint32_t db = __float_as_uint(depths[idx]);
            if (depth_code(db, SPLIT_BITS) > near_code) {
                r = rects[idx];
                entry = (uint32_t)idx;
                if (code_bits) entry |= depth_code(db, (uint32_t)code_bits) << (32 - code_bits);
            }
        }
        const int ya = max((int)r.y, y0), yb = min((int)r.w, y1 + 1);
        for_each_tile<16>(r.z > r.x && yb > ya, r.x, ya, r.z, yb, (int)entry, [&](int x, int y, int id) {
            const int t = y * gx + x;
            if (t >= t0 && t < t1 && tile_state[t] != 0xffffffffu)
                bucket_ids[tile_offset[t] + tile_near[t] + atomicAdd(&far_cursor[t], 1u)] = (uint32_t)id;
        });
    }
}

// Threshold selection, two small kernels.  (1) histogram of the visible Gaussians' SPLIT_BITS-bit depth codes, each weighted by the
// number of tiles it touches (= its instances), per workgroup in LDS, then added to the global bins; (2) one workgroup scans the
// bins and takes the first code at which the running instance count reaches near_per_tile * tiles -- or switches the split off
// when the frame is not dense (fewer than dense_avg instances per tile, unless forced) or the threshold would keep everything.
__global__ void __launch_bounds__(1024) split_hist_kernel(int P, const float* __restrict__ depths, const uint32_t* __restrict__ tiles_touched,
                                                          uint32_t* __restrict__ code_hist) {
    __shared__ uint32_t h[SPLIT_BINS];
    for (uint32_t i = threadIdx.x; i < SPLIT_BINS; i += 1024) h[i] = 0;
    __syncthreads();
    // four elements per thread and trip, their loads issued together (one element per trip was one memory round trip per element:
    // 78 us for the 10 M Gaussians of config 5 on 128 workgroups)
    const int stride = gridDim.x * 1024;
    for (int i0 = blockIdx.x * 1024 + threadIdx.x; i0 < P; i0 += 4 * stride) {
        uint32_t w[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * stride;
            w[u] = i < P ? tiles_touched[i] : 0u;
            d[u] = i < P ? __float_as_uint(depths[i]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (w[u]) atomicAdd(&h[depth_code(d[u], SPLIT_BITS)], w[u]);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < SPLIT_BINS; i += 1024)
        if (h[i]) atomicAdd(&code_hist[i], h[i]);
}

__global__ void __launch_bounds__(1024) split_pick_kernel(const uint32_t* __restrict__ code_hist, uint32_t tiles, uint32_t near_per_tile,
                                                          uint32_t dense_avg, int force, SplitState* __restrict__ split) {
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t pick;
    constexpr int PER = SPLIT_BINS / 1024;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v[PER];
    unsigned long long local = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { v[k] = code_hist[tid * PER + k]; local += v[k]; }
    unsigned long long incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    if (tid == 0) pick = SPLIT_OFF;
    __syncthreads();
    unsigned long long base = 0, total = 0;
    for (uint32_t w = 0; w < 16; w++) {
        if (w < wave) base += wsum[w];
        total += wsum[w];
    }
    const unsigned long long want = (unsigned long long)near_per_tile * tiles;
    const bool on = want < total && (force || total >= (unsigned long long)dense_avg * tiles);
    if (on) {
        unsigned long long run = base + incl - local;  // instances of all smaller codes
#pragma unroll
        for (int k = 0; k < PER; k++) {
            // the first code whose inclusive count reaches `want`: exactly one (tid, k) satisfies this
            if (run < want && run + v[k] >= want) pick = tid * PER + k;
            run += v[k];
        }
    }
    __syncthreads();
    if (tid == 0) {
        // a threshold at the last code keeps everything near: nothing to gain
        split->near_code = (pick != SPLIT_OFF && pick + 1 < SPLIT_BINS) ? pick : SPLIT_OFF;
        split->need_far = 0u;
        split->far_tiles = 0u;
        split->aim = near_per_tile;
    }
}

hipError_t launch_split_threshold(int P, const GeometryState& g, const ImageState& img, int tiles, bool force, uint32_t near_per_tile,
                                  hipStream_t stream) {
    hipError_t e = hipMemsetAsync(img.code_hist, 0, SPLIT_BINS * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(img.far_cursor, 0, (size_t)tiles * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const int blocks = P >= (1 << 22) ? 512 : P >= (1 << 20) ? 256 : max(1, (P + 8191) / 8192);  // (each flushes its LDS bins with global atomics)
    hipLaunchKernelGGL(split_hist_kernel, dim3(blocks), dim3(1024), 0, stream, P, g.depths, g.tiles_touched, img.code_hist);
    hipLaunchKernelGGL(split_pick_kernel, dim3(1), dim3(1024), 0, stream, img.code_hist, (uint32_t)tiles, near_per_tile, SPLIT_DENSE_AVG, force ? 1 : 0,
                       img.split);
    return hipGetLastError();
}

hipError_t launch_tile_scatter_far(int P, const GeometryState& g, const ImageState& img, const BinningState& b, int gx, int tiles, int code_bits,
                                   const BinStats* guard, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    if (g.band_list != nullptr)
        hipLaunchKernelGGL(tile_scatter_far_kernel<true>, dim3(BIN_CHUNKS * 8), dim3(256), 0, stream, P, g.rects, g.depths, img.tile_offset,
                           img.tile_near, img.tile_state, img.far_cursor, g.band_list, g.band_cnt, b.bucket_ids, gx, tiles, code_bits, img.split, guard);
    else
        hipLaunchKernelGGL(tile_scatter_far_kernel<false>, dim3(BIN_CHUNKS * 8), dim3(256), 0, stream, P, g.rects, g.depths, img.tile_offset,
                           img.tile_near, img.tile_state, img.far_cursor, (const uint16_t*)nullptr, (const uint32_t*)nullptr, b.bucket_ids, gx,
                           tiles, code_bits, img.split, guard);
    return hipGetLastError();
}

// ---- per-tile sort ----------------------------------------------------------------------------------------------------
// Bitonic sort of a tile's bucket.  Keys are (depth_bits << 32 | gaussian id), built while loading (the id comes from the
// bucket, the depth from the 4 B/Gaussian depth array, which stays in L2); padded to a power of two with ~0.
//
// The whole network runs on keys held IN REGISTERS, E consecutive keys per thread (256 threads, E = 4 for tiles of up to
// 1024 instances, 8 / 16 / 32 for longer lists up to 8192):
//   distance < E          : compare-exchange between a thread's own registers;
//   distance E .. 32E     : the partner keys sit in lane  l ^ (j/E)  of the same wave -> DPP quad_perm / row_ror,
//                           ds_swizzle or ds_bpermute (crossbar only: no LDS memory traffic, no bank conflicts, no barrier);
//   distance 64E, 128E    : the partner is in another wave -> one round trip through LDS with workgroup barriers
//                           (3 of the 55 stages of a 1024-key sort).
// A first version kept the keys in LDS for every stage; rocprofv3 showed its LDS pipe ~80 % busy (37 % of that bank
// conflicts on the 64-bit accesses).
// The host knows the longest list (mailbox) and launches the instantiation whose EMAX covers it; inside, every workgroup
// takes the smallest E that holds its own tile, so the few long tiles sort while the short ones do (a second launch for
// them cost 23 us on the headline scene: one 2048-key network is a 20 us dependency chain).  EMAX = 32 would cost the
// short tiles their occupancy (118 VGPRs, 64 KB LDS), so lists of 4097..8192 get their own launch after the <16> one.
template <int EMAX>
__global__ void __launch_bounds__(256) tile_sort_kernel(const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ bucket_ids,
                                                        const float* __restrict__ depths, uint32_t* __restrict__ point_list, uint32_t n_min,
                                                        uint32_t n_max, uint32_t id_mask, const BinStats* __restrict__ guard) {
    extern __shared__ uint64_t skeys[];
    if (guard && guard->spec_fail) return;
    const int tile = blockIdx.x;
    const uint32_t begin = tile_offset[tile];
    const uint32_t n = tile_offset[tile + 1] - begin;
    if (n < n_min || n > n_max) return;  // n == 0, or a tile the other launch takes care of
    if constexpr (EMAX == 32) {
        tile_sort_body<32>(skeys, n, bucket_ids + begin, depths, point_list + begin, id_mask);
    } else {
        // the smallest network that holds the list: its keys are then spread over all four waves (a 450-key list in the 1024-key network
        // leaves waves 2 and 3 sorting padding while waves 0 and 1 carry the whole dependency chain: config 2's sort 0.049 -> 0.041 ms)
        // (only in the launch for frames whose longest list is <= 1024: a frame with longer lists has hardly any this short, and the third
        //  network in that kernel's code cost the headline frame 3 us)
        if (EMAX == 4 && n <= 512) tile_sort_body<2>(skeys, n, bucket_ids + begin, depths, point_list + begin, id_mask);
        else if (EMAX == 4 || n <= 1024) tile_sort_body<4>(skeys, n, bucket_ids + begin, depths, point_list + begin, id_mask);
        else if (EMAX == 8 || n <= 2048) tile_sort_body<(EMAX >= 8 ? 8 : 4)>(skeys, n, bucket_ids + begin, depths, point_list + begin, id_mask);
        else tile_sort_body<(EMAX >= 16 ? 16 : 4)>(skeys, n, bucket_ids + begin, depths, point_list + begin, id_mask);
    }
}

// ---- lazy sort, first round ----------------------------------------------------------------------------------------------
// Tiles listing at most min_len instances are sorted in full.  Of a longer list only a front of about `target` depth-nearest
// instances is extracted (wg_sort.h: extract_front) and sorted into point_list; seg_end[tile] tells the forward pass how far
// the list is in order.  The tile's bucket is left as it is.

// One workgroup per tile: extract (long list) or take (short list) the ids, sort them in registers, write point_list.
__global__ void __launch_bounds__(256) tile_front_sort_kernel(const uint32_t* __restrict__ tile_offset, const uint32_t* __restrict__ bucket_ids,
                                                              const float* __restrict__ depths, uint32_t* __restrict__ point_list,
                                                              uint32_t* __restrict__ seg_end, uint32_t min_len, uint32_t target, uint32_t cap,
                                                              uint32_t id_mask, const uint32_t* __restrict__ tile_near, const BinStats* __restrict__ guard) {
    // the selection scratch (12 KB) and the sort's cross-wave exchange buffer (16 KB) are never live together: one 16 KB block,
    // which lets 8 workgroups share a CU instead of 5 (the kernel is a chain of dependent phases, latency-bound)
    __shared__ uint64_t smem[256 * 8];
    static_assert(sizeof(SelectScratch) <= sizeof(uint64_t) * 256 * 8, "selection scratch must fit the exchange buffer");
    SelectScratch& sc = *reinterpret_cast<SelectScratch*>(smem);
    uint64_t* skeys = smem;
    if (guard && guard->spec_fail) return;
    const int tile = blockIdx.x;
    const uint32_t begin = tile_offset[tile];
    // near / far split: only the near instances, the first tile_near[tile] entries of the bucket, exist at this point
    const uint32_t n = tile_near ? tile_near[tile] : tile_offset[tile + 1] - begin;
    if (n == 0) {
        if (threadIdx.x == 0) seg_end[tile] = 0;
        return;
    }
    const uint32_t* src = bucket_ids + begin;
    uint32_t len = n;
    if (n > min_len) {  // workgroup-uniform
        len = extract_front(bucket_ids + begin, n, depths, 0ull, n, target, cap, id_mask, sc);
        src = sc.ids;  // plain ids
    }
    // (the smallest network that holds the front: a near bag of ~450 instances on the 512-key network keeps all four waves busy)
    if (len <= 512) tile_sort_body<2>(skeys, len, src, depths, point_list + begin, id_mask);
    else if (len <= 1024) tile_sort_body<4>(skeys, len, src, depths, point_list + begin, id_mask);
    else tile_sort_body<8>(skeys, len, src, depths, point_list + begin, id_mask);
    if (threadIdx.x == 0) seg_end[tile] = len;
}

// ---- launch order of the render kernels ---------------------------------------------------------------------------
// The render kernels run one wave per tile and all ~8k waves are resident at once (<= 8 per SIMD), so the kernel ends
// when the SIMD with the largest SUM of tile costs ends.  Dealing the tiles of each XCD band in descending cost order
// gives every SIMD one tile of each size class.  One workgroup per band orders its tiles.  Pure scheduling: results do not
// depend on it.  Used for the backward kernel, whose
// per-tile cost (tile_last, the walked length) is known exactly from the forward pass: 0.752 -> 0.660 ms.  The forward
// kernel's only predictor within the frame, the list length, did not help (its walked fraction is what varies): it is ordered by the tile costs
// of an EARLIER frame of the same camera (forward_order_kernel below).
// The order only has to be roughly descending, so it is a 256-bin counting sort (bin = cost scaled by the band's maximum; 5
// barriers) rather than a comparison sort of the band's ~1000 keys (55 barriers: 14 us of pure latency in front of the backward
// kernel).
// Blocks 8 and up (when asked for) clear the backward pass's gradient records: the clear and the ordering are both needed in front of
// the per-tile kernel and depend on nothing of each other, so they share a launch instead of queueing as a fill kernel + this one.
__global__ void __launch_bounds__(1024) tile_order_kernel(const uint32_t* __restrict__ cost, const uint2* __restrict__ ranges,
                                                          uint32_t* __restrict__ order, int tiles, float4* __restrict__ clear, size_t clear_vec4,
                                                          uint32_t period) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wmax[16];
    if (blockIdx.x >= 8) {
        const size_t stride = (size_t)(gridDim.x - 8) * 1024;
        for (size_t i = (size_t)(blockIdx.x - 8) * 1024 + threadIdx.x; i < clear_vec4; i += stride) clear[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int q = tiles >> 3, rem = tiles & 7, x = blockIdx.x;
    const uint32_t start = x * q + min(x, rem), cnt = q + (x < rem ? 1 : 0);
    order_band([&](uint32_t i) { return cost ? cost[start + i] : (ranges[start + i].y - ranges[start + i].x); }, start, cnt, order, period, hist, wmax);
}

hipError_t launch_tile_order(const uint32_t* cost_or_null, const uint2* ranges_or_null, uint32_t* order, int tiles, float* clear, size_t clear_floats,
                             int period, hipStream_t stream) {
    if (tiles <= 0) return hipSuccess;
    // clear: 16-byte aligned (the gradient records: 12 floats per Gaussian, + P floats in a two-colour call, in a 256-byte aligned array)
    const size_t vec4 = clear ? (clear_floats + 3) / 4 : 0;   // (rounded UP: 13 floats per Gaussian in a two-colour call; the array is carved to a whole float4)
    const unsigned clear_blocks = vec4 ? (unsigned)std::min<size_t>(2048, (vec4 + 4095) / 4096) : 0u;
    hipLaunchKernelGGL(tile_order_kernel, dim3(8 + clear_blocks), dim3(1024), 0, stream, cost_or_null, ranges_or_null, order, tiles,
                       reinterpret_cast<float4*>(clear), vec4, (uint32_t)std::max(period, 0));
    return hipGetLastError();
}

// Dynamic LDS above 64 KiB needs the function attribute raised.  Done once per (device, kernel, size high-water mark): the
// call takes a driver lock and was seen to stall the launching thread for milliseconds when issued on every frame.
static hipError_t ensure_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{dev, fn}];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

hipError_t launch_tile_count(int P, const GeometryState& g, const ImageState& img, int gx, int tiles, bool split, bool box, bool fused_scan,
                             hipStream_t stream) {
    const size_t lds = (size_t)tiles * sizeof(uint32_t);
    const SplitState* sp = split ? img.split : nullptr;
    const bool bx = box && gx <= BIN_THREADS, ls = g.band_list != nullptr;
    hipError_t e = hipSuccess;
#define WG_COUNT(B, L)                                                                                                                         \
    do {                                                                                                                                       \
        e = ensure_lds(reinterpret_cast<const void*>(tile_count_kernel<B, L>), lds);                                                           \
        if (e != hipSuccess) return e;                                                                                                         \
        hipLaunchKernelGGL((tile_count_kernel<B, L>), dim3(BIN_CHUNKS), dim3(BIN_THREADS), lds, stream, P, g.rects, img.chunk_hist, g.band_list, \
                           g.band_cnt, gx, tiles, g.depths, sp, img.scan_ticket);                                                              \
    } while (0)
    if (bx && ls) WG_COUNT(true, true);
    else if (bx) WG_COUNT(true, false);
    else if (ls) WG_COUNT(false, true);
    else WG_COUNT(false, false);
#undef WG_COUNT
    e = hipGetLastError();
    if (e != hipSuccess || fused_scan) return e;  // fused: launch_tile_scan runs the column scan too
    hipLaunchKernelGGL(chunk_scan_kernel, dim3((tiles + 63) / 64), dim3(64 * SCAN_WAVES), 0, stream, img.chunk_hist, img.tile_count, tiles,
                       split ? img.tile_near : (uint32_t*)nullptr, sp);
    return hipGetLastError();
}

hipError_t launch_tile_scan(const ImageState& img, int tiles, HostMailbox* mailbox_dev, uint32_t seq, bool split, const SpecLimits& spec,
                            bool fused_scan, const FwdOrderArgs& fo, hipStream_t stream) {
    const SplitState* sp = split ? img.split : nullptr;
    if (fused_scan) {
        uint32_t* near = split ? img.tile_near : (uint32_t*)nullptr;
        if (tiles <= 8 * 1024)
            hipLaunchKernelGGL(chunk_tile_scan_kernel<8>, dim3((tiles + 63) / 64), dim3(64 * SCAN_WAVES), 0, stream, img.chunk_hist, img.tile_count, tiles,
                               near, sp, img.scan_ticket, img.tile_offset, img.ranges, img.stats, mailbox_dev, seq, spec);
        else
            hipLaunchKernelGGL((chunk_tile_scan_kernel<BIN_MAX_TILES / 1024>), dim3((tiles + 63) / 64), dim3(64 * SCAN_WAVES), 0, stream, img.chunk_hist,
                               img.tile_count, tiles, near, sp, img.scan_ticket, img.tile_offset, img.ranges, img.stats, mailbox_dev, seq, spec);
        return hipGetLastError();
    }
    // BIN_MAX_TILES / 1024 = 36 tiles per thread at most; 8 covers 1080p (8160 tiles)
    if (tiles <= 8 * 1024)
        hipLaunchKernelGGL(tile_scan_kernel<8>, dim3(fo.table ? 9 : 1), dim3(1024), 0, stream, img.tile_count, img.tile_offset, img.ranges, img.stats, tiles,
                           mailbox_dev, seq, sp, spec, fo);
    else
        hipLaunchKernelGGL((tile_scan_kernel<BIN_MAX_TILES / 1024>), dim3(fo.table ? 9 : 1), dim3(1024), 0, stream, img.tile_count, img.tile_offset, img.ranges,
                           img.stats, tiles, mailbox_dev, seq, sp, spec, fo);
    return hipGetLastError();
}


hipError_t launch_tile_scatter(int P, const GeometryState& g, const ImageState& img, const BinningState& b, int gx, int tiles,
                               uint32_t num_rendered, int code_bits, int g_staged_scatter, int g_staged_cap, bool split, const BinStats* guard,
                               hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    const SplitState* sp = split ? img.split : nullptr;
    const uint32_t* last_prefix = split ? img.tile_near : img.tile_count;  // what follows the last chunk's prefix in a tile's column
    // staged scatter for long lists: G chunks per workgroup such that an average share fits the staging area with headroom
    const bool want = g_staged_scatter == 1 || (g_staged_scatter < 0 && tiles > 0 && num_rendered / (uint32_t)tiles >= 1500u);
    if (want) {
        const int nbmax = tiles / 8 + 1;
        const size_t fixed = ((size_t)2 * nbmax + 16) * sizeof(uint32_t);
        // two 1024-thread workgroups per CU; a share (chunks x band) larger than the staging area is emitted in several passes
        // (one workgroup per CU with twice the room was measured slower: 0.32 vs 0.21 ms at 8450 per tile)
        const double share1 = (double)num_rendered / (double)(BIN_CHUNKS * 8);
        const size_t budget = 78 * 1024;
        if (fixed + 4096 * sizeof(uint32_t) <= budget) {
            uint32_t cap = (uint32_t)((budget - fixed) / sizeof(uint32_t));
            if (g_staged_cap > 0 && (uint32_t)g_staged_cap < cap) cap = (uint32_t)g_staged_cap;
            int G = 4;
            while (G > 1 && share1 * G * 1.5 > (double)cap) G >>= 1;
            const size_t lds = fixed + (size_t)cap * sizeof(uint32_t);
            hipError_t e = ensure_lds(reinterpret_cast<const void*>(tile_scatter_staged_kernel<false>), lds);
            if (e == hipSuccess) e = ensure_lds(reinterpret_cast<const void*>(tile_scatter_staged_kernel<true>), lds);
            if (e != hipSuccess) return e;
            // the band lists (large P) replace one scan of the chunk per band and pass by a dependent list -> rectangle load
            // chain: worth it when a share takes several passes, or when there are few instances per Gaussian to amortise a scan
            const bool use_lists = g.band_list != nullptr && (share1 * G > (double)cap || (double)num_rendered < 10.0 * (double)P);
            if (use_lists)
                hipLaunchKernelGGL(tile_scatter_staged_kernel<true>, dim3(BIN_CHUNKS / G * 8), dim3(1024), lds, stream, P, g.rects, g.depths,
                                   img.tile_offset, last_prefix, img.chunk_hist, g.band_list, g.band_cnt, b.bucket_ids, gx, tiles, G, nbmax,
                                   cap, code_bits, sp, guard);
            else
                hipLaunchKernelGGL(tile_scatter_staged_kernel<false>, dim3(BIN_CHUNKS / G * 8), dim3(1024), lds, stream, P, g.rects, g.depths,
                                   img.tile_offset, last_prefix, img.chunk_hist, (const uint16_t*)nullptr, (const uint32_t*)nullptr,
                                   b.bucket_ids, gx, tiles, G, nbmax, cap, code_bits, sp, guard);
            return hipGetLastError();
        }
    }
    const size_t lds = (size_t)(tiles / 8 + 1) * sizeof(uint32_t);
    hipError_t e = ensure_lds(reinterpret_cast<const void*>(tile_scatter_kernel<false>), lds);
    if (e == hipSuccess) e = ensure_lds(reinterpret_cast<const void*>(tile_scatter_kernel<true>), lds);
    if (e != hipSuccess) return e;
    if (g.band_list != nullptr)
        hipLaunchKernelGGL(tile_scatter_kernel<true>, dim3(BIN_CHUNKS * 8), dim3(256), lds, stream, P, g.rects, g.depths, img.tile_offset,
                           img.chunk_hist, g.band_list, g.band_cnt, b.bucket_ids, gx, tiles, code_bits, sp, guard);
    else
        hipLaunchKernelGGL(tile_scatter_kernel<false>, dim3(BIN_CHUNKS * 8), dim3(256), lds, stream, P, g.rects, g.depths, img.tile_offset,
                           img.chunk_hist, (const uint16_t*)nullptr, (const uint32_t*)nullptr, b.bucket_ids, gx, tiles, code_bits, sp, guard);
    return hipGetLastError();
}

template <int E>
static hipError_t launch_tile_sort_e(const ImageState& img, const BinningState& b, const GeometryState& g, int tiles, uint32_t n_min,
                                     uint32_t n_max, const BinStats* guard, hipStream_t stream) {
    const size_t lds = (size_t)256 * E * sizeof(uint64_t);  // exchange buffer of the cross-wave stages
    hipError_t e = ensure_lds(reinterpret_cast<const void*>(tile_sort_kernel<E>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tile_sort_kernel<E>, dim3(tiles), dim3(256), lds, stream, img.tile_offset, b.bucket_ids, g.depths, b.point_list, n_min,
                       n_max, 0xffffffffu, guard);
    return hipGetLastError();
}

hipError_t launch_tile_sort_lazy(const ImageState& img, const BinningState& b, const GeometryState& g, int tiles, int code_bits,
                                 const LazyConfig& g_lazy, bool split, const BinStats* guard, hipStream_t stream) {
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_front_sort_kernel, dim3(tiles), dim3(256), 0, stream, img.tile_offset, b.bucket_ids, g.depths, b.point_list,
                       img.seg_end, g_lazy.min_len, g_lazy.target, g_lazy.cap, code_bits ? (1u << (32 - code_bits)) - 1u : 0xffffffffu,
                       split ? img.tile_near : (const uint32_t*)nullptr, guard);
    return hipGetLastError();
}

hipError_t launch_tile_sort(const ImageState& img, const BinningState& b, const GeometryState& g, int tiles, uint32_t max_count,
                            const BinStats* guard, hipStream_t stream) {
    if (tiles <= 0 || max_count == 0) return hipSuccess;
    if (max_count <= 1024) return launch_tile_sort_e<4>(img, b, g, tiles, 1u, 1024u, guard, stream);
    if (max_count <= 2048) return launch_tile_sort_e<8>(img, b, g, tiles, 1u, 2048u, guard, stream);
    hipError_t e = launch_tile_sort_e<16>(img, b, g, tiles, 1u, 4096u, guard, stream);
    if (e != hipSuccess || max_count <= 4096) return e;
    return launch_tile_sort_e<32>(img, b, g, tiles, 4097u, TILE_SORT_MAX, guard, stream);
}

}  // namespace wg
