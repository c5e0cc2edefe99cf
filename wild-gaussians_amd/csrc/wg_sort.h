// Register-resident bitonic sort of a tile's instance keys and the sampled front selection of the lazy sort.
// Device functions shared by binning.hip (tile_sort_kernel, tile_select_kernel) and render_fwd.hip (fix-up kernel); all of
// them are WORKGROUP-level: every one of the 256 threads must call them (they contain __syncthreads()).
#pragma once
#include "wg_common.h"

namespace wg {

__device__ __forceinline__ void bitonic_ce(uint64_t* skeys, uint32_t t, uint32_t j, uint32_t k) {
    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const uint32_t l = i | j;
    const uint64_t a = skeys[i], b = skeys[l];
    const bool ascending = (i & k) == 0;
    if ((a > b) == ascending) {
        skeys[i] = b;
        skeys[l] = a;
    }
}

template <int D>  // value of lane (l ^ D), D in {1, 2, 4, 8, 16, 32}
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror 8
    else if constexpr (D == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                     // xor 4 (bit mode)
    else if constexpr (D == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                    // xor 16
    else return (uint32_t)__shfl_xor((int)v, 32);
}

template <int E>
struct SortCtx {
    uint64_t key[E];  // E consecutive keys of the tile's padded array
    uint32_t t;       // thread index in the workgroup
    uint32_t gidx;    // index of key[0] in the padded array (= E * t: one chunk of 256*E keys covers the tile)
    uint64_t* xchg;   // LDS exchange buffer (256*E keys) for the cross-wave stages
};

__device__ __forceinline__ void keep(uint64_t& mine, uint64_t other, bool keep_min) {
    const bool other_less = other < mine;
    mine = (other_less == keep_min) ? other : mine;
}

template <int E, int K, int J>
__device__ __forceinline__ void sort_stage(SortCtx<E>& c) {
    if constexpr (J < E) {  // both keys of every pair live in this thread's registers
#pragma unroll
        for (int r = 0; r < E; r++) {
            if ((r & J) == 0) {
                const bool asc = ((c.gidx + r) & K) == 0;
                uint64_t& a = c.key[r];
                uint64_t& b = c.key[r | J];
                if ((a > b) == asc) {
                    const uint64_t tmp = a;
                    a = b;
                    b = tmp;
                }
            }
        }
    } else {
        const bool asc = (c.gidx & K) == 0;
        const bool lower = (c.t & (J / E)) == 0;
        const bool keep_min = lower == asc;
        if constexpr (J < 64 * E) {  // the partner keys sit in lane l ^ (J/E) of this wave
#pragma unroll
            for (int r = 0; r < E; r++) {
                const uint32_t lo = lane_xor<J / E>((uint32_t)c.key[r]);
                const uint32_t hi = lane_xor<J / E>((uint32_t)(c.key[r] >> 32));
                keep(c.key[r], ((uint64_t)hi << 32) | lo, keep_min);
            }
        } else {  // the partner keys sit in another wave: one round trip through LDS
            uint64_t* mine = c.xchg + E * c.t;
#pragma unroll
            for (int r = 0; r < E; r++) mine[r] = c.key[r];
            __syncthreads();
            const uint64_t* theirs = c.xchg + E * (c.t ^ (J / E));
            uint64_t other[E];
#pragma unroll
            for (int r = 0; r < E; r++) other[r] = theirs[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < E; r++) keep(c.key[r], other[r], keep_min);
        }
    }
}

template <int E, int K, int J>
__device__ __forceinline__ void sort_stages(SortCtx<E>& c) {
    sort_stage<E, K, J>(c);
    if constexpr (J > 1) sort_stages<E, K, J / 2>(c);
}

template <int E, int K>
__device__ __forceinline__ void sort_levels(SortCtx<E>& c, uint32_t kmax) {
    if constexpr (K > 2) sort_levels<E, K / 2>(c, kmax);
    if (K <= kmax) sort_stages<E, K, K / 2>(c);  // kmax is workgroup-uniform
}

// One workgroup per tile; E keys per thread sort a tile of up to 256*E instances.
// src and dst may be the same array (every key is in a register before the first store).
template <int E>
__device__ __forceinline__ void tile_sort_body(uint64_t* skeys, uint32_t begin, uint32_t n, const uint32_t* bucket_ids,
                                               const float* __restrict__ depths, uint32_t* point_list) {
    uint32_t np2 = E;  // at least one key group
    while (np2 < n) np2 <<= 1;
    SortCtx<E> c;
    c.t = threadIdx.x;
    c.gidx = E * threadIdx.x;
    c.xchg = skeys;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t i = c.gidx + r;
        uint64_t key = ~0ull;
        if (i < n) {
            const uint32_t id = bucket_ids[begin + i];
            key = ((uint64_t)__float_as_uint(depths[id]) << 32) | id;
        }
        c.key[r] = key;
    }
    sort_levels<E, 256 * E>(c, np2);
    __syncthreads();  // in-place use: nobody stores before everybody has loaded
#pragma unroll
    for (int r = 0; r < E; r++)
        if (c.gidx + r < n) point_list[begin + c.gidx + r] = (uint32_t)c.key[r];
}


// ---- lazy sort: front selection ------------------------------------------------------------------------------------
// A long list is not sorted in full: the forward pass stops at the first few hundred entries of a dense tile (3.6 % of
// the instances at 5000 per tile), so only a depth-nearest FRONT of the list is split off and sorted; the rest stays an
// unsorted bag until a tile turns out to need more (render_fwd.hip: fix-up kernel).
__device__ __forceinline__ uint64_t depth_key(const float* __restrict__ depths, uint32_t id) {
    return ((uint64_t)__float_as_uint(depths[id]) << 32) | id;
}

struct SelectScratch {   // LDS
    uint64_t sample[256];
    uint64_t sorted[256];
    uint32_t wsum[8];
    uint32_t totals[2];
};

// One partition pass over bag[0..m): ids whose key <= thr are appended to front[0..) (any order), the others are compacted
// in place to the END of the bag (bag[F..m) afterwards).  Returns F.  The bag is read in chunks from its end, so every
// in-place store lands on entries that are already in registers.
__device__ __forceinline__ uint32_t partition_pass(uint32_t* bag, uint32_t m, const float* __restrict__ depths, uint64_t thr,
                                                   uint32_t* front, SelectScratch& sc) {
    constexpr uint32_t PER = 8, CH = 256 * PER;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t f_head = 0, b_head = m;
    for (uint32_t chunk_end = m; chunk_end > 0; chunk_end -= min(chunk_end, CH)) {
        const uint32_t chunk_begin = chunk_end > CH ? chunk_end - CH : 0;
        uint32_t id[PER];
        uint32_t isf = 0, nf = 0, nb = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = chunk_begin + k * 256 + tid;
            if (i < chunk_end) {
                id[k] = bag[i];
                if (depth_key(depths, id[k]) <= thr) { isf |= 1u << k; nf++; }
                else nb++;
            }
        }
        // exclusive scan of (nf, nb) over the workgroup, packed 16:16
        const uint32_t packed = nf | (nb << 16);
        uint32_t incl = packed;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        __syncthreads();  // every thread holds its entries; the previous chunk's use of wsum is over
        if (lane == 63) sc.wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (uint32_t w = 0; w < 4; w++) {
            if (w < wave) base += sc.wsum[w];
            total += sc.wsum[w];
        }
        const uint32_t excl = base + incl - packed;
        uint32_t fpos = f_head + (excl & 0xffffu), bpos = b_head - (excl >> 16);
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = chunk_begin + k * 256 + tid;
            if (i < chunk_end) {
                if ((isf >> k) & 1u) front[fpos++] = id[k];
                else bag[--bpos] = id[k];
            }
        }
        f_head += total & 0xffffu;
        b_head -= total >> 16;
    }
    __syncthreads();
    return f_head;
}

// Splits about `target` depth-nearest ids off bag[0..m) into front[0..F), 1 <= F <= cap, leaving the others in
// bag[F..m).  The threshold is a quantile of 256 evenly spaced samples; if the split comes out larger than cap (it is tightly
// concentrated around target, so this is the pathological case) the front is poured back and the target halved, down to the
// single nearest id, which always terminates.
__device__ __forceinline__ uint32_t select_front(uint32_t* bag, uint32_t m, const float* __restrict__ depths, uint32_t* front,
                                                 uint32_t target, uint32_t cap, SelectScratch& sc) {
    const uint32_t tid = threadIdx.x;
    for (;;) {
        uint64_t thr;
        if (target > 0) {
            const uint64_t mine = depth_key(depths, bag[(uint32_t)(((uint64_t)tid * m) >> 8)]);
            __syncthreads();
            sc.sample[tid] = mine;
            __syncthreads();
            uint32_t rank = 0;  // ties (m < 256 samples the same entry twice) are broken by the thread index
            for (uint32_t j = 0; j < 256; j++) rank += (sc.sample[j] < mine || (sc.sample[j] == mine && j < tid)) ? 1u : 0u;
            sc.sorted[rank] = mine;
            __syncthreads();
            const uint32_t q = min(254u, (uint32_t)(((uint64_t)target << 8) / m));
            thr = sc.sorted[q];
        } else {  // last resort: the single nearest entry
            uint64_t best = ~0ull;
            for (uint32_t i = tid; i < m; i += 256) best = min(best, depth_key(depths, bag[i]));
            __syncthreads();
            sc.sample[tid] = best;
            __syncthreads();
            for (uint32_t j = 0; j < 256; j++) best = min(best, sc.sample[j]);
            thr = best;
        }
        const uint32_t F = partition_pass(bag, m, depths, thr, front, sc);
        if (F <= cap) return F;
        for (uint32_t i = tid; i < F; i += 256) bag[i] = front[i];  // pour the front back into the gap bag[0..F)
        __syncthreads();
        target >>= 1;
    }
}

}  // namespace wg
