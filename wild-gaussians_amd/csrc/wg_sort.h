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
            // key r of thread t at xchg[256 r + t]: consecutive lanes on consecutive 8-byte words, for the writes and -- the partner
            // thread t ^ (J/E) differs from t in a wave-index bit only -- for the reads alike.  (Round 2 kept a thread's E keys
            // together, xchg[E t + r]: a 32-byte lane stride, 9.7 M LDS bank-conflict cycles per launch at the headline scene.)
#ifndef WG_SORT_XCHG_TRANSPOSED
#define WG_SORT_XCHG_TRANSPOSED 1
#endif
            constexpr uint32_t RS = WG_SORT_XCHG_TRANSPOSED ? 256u : 1u, TS = WG_SORT_XCHG_TRANSPOSED ? 1u : (uint32_t)E;
            uint64_t* mine = c.xchg + TS * c.t;
#pragma unroll
            for (int r = 0; r < E; r++) mine[RS * r] = c.key[r];
            __syncthreads();
            const uint64_t* theirs = c.xchg + TS * (c.t ^ (J / E));
            uint64_t other[E];
#pragma unroll
            for (int r = 0; r < E; r++) other[r] = theirs[RS * r];
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

// One workgroup sorts n <= 256*E instance ids (E keys per thread) read from src (global or LDS) and writes them, ordered by
// (depth, id), to dst.  src and dst may overlap (every key is in a register before the first store).
// id_mask strips the coarse depth code bucket entries may carry above the id (see depth_code below).
template <int E>
__device__ __forceinline__ void tile_sort_body(uint64_t* skeys, uint32_t n, const uint32_t* src, const float* __restrict__ depths,
                                               uint32_t* dst, uint32_t id_mask) {
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
            const uint32_t id = src[i] & id_mask;
            key = ((uint64_t)__float_as_uint(depths[id]) << 32) | id;
        }
        c.key[r] = key;
    }
    __syncthreads();  // src may share LDS with the exchange buffer (skeys): every key is in a register before it is reused
    sort_levels<E, 256 * E>(c, np2);
    __syncthreads();  // overlapping src / dst: nobody stores before everybody has loaded
#pragma unroll
    for (int r = 0; r < E; r++)
        if (c.gidx + r < n) dst[c.gidx + r] = (uint32_t)c.key[r];
}

// ---- lazy sort: front selection ------------------------------------------------------------------------------------
// A long list is not sorted in full: the forward pass stops at the first few hundred entries of a dense tile (3.6 % of
// the instances at 5000 per tile), so only a depth-nearest FRONT of the list is extracted and sorted; the tile's bucket
// stays an unsorted bag, and a further front (the keys just above the last one taken) is extracted from it only if a tile
// turns out to need more (render_fwd.hip: fix-up kernel).  Nothing is moved inside the bag: "already taken" is simply
// "key <= the last key of the sorted part".
__device__ __forceinline__ uint64_t depth_key(const float* __restrict__ depths, uint32_t id) {
    return ((uint64_t)__float_as_uint(depths[id]) << 32) | id;
}

// Coarse, monotone code of a depth (> 0.2, the near cull) in `cb` bits: 2^(cb-4) steps per octave (8 bits: 1/16 octave per
// step), saturating beyond 0.2 * 2^16.  When the Gaussian ids leave room (at most 2^24 of them: 8 bits; 2^20: 12 bits) the
// scatter kernels put it above the id in every bucket entry, and the extraction pass below decides most entries from the code
// alone: the exact depth (a 4-byte gather that costs a 64-byte sector, and misses the L2 once the depth array outgrows it) is
// fetched only for entries whose code equals that of a bound.
__device__ __forceinline__ uint32_t depth_code(uint32_t depth_bits, uint32_t cb) {
    const uint32_t near_bits = 0x3E4CCCCDu;  // 0.2f
    if (cb == 0u || depth_bits <= near_bits) return 0u;
    return min((1u << cb) - 1u, (depth_bits - near_bits) >> (27u - cb));
}
__device__ __forceinline__ uint32_t code_bits_of(uint32_t id_mask) { return 32u - (uint32_t)__builtin_popcount(id_mask); }  // 0: plain ids
constexpr int MIN_ID_BITS = 20, MAX_ID_BITS_CODED = 24;  // code widths 12 .. 8

constexpr uint32_t FRONT_CAP = 2048;  // the 8-keys-per-thread network

struct SelectScratch {  // LDS
    uint64_t sample[256];
    uint64_t sorted[256];
    uint32_t ids[FRONT_CAP];
    uint32_t count;
    uint32_t nvalid;
};

// One pass over the bag: the ids with lo < key <= thr are appended to sc.ids (any order, at most cap kept).  Returns how
// many there are (possibly more than cap: the caller then lowers thr and repeats).
__device__ __forceinline__ uint32_t extract_pass(const uint32_t* __restrict__ bag, uint32_t n, const float* __restrict__ depths, uint64_t lo,
                                                 uint64_t thr, uint32_t cap, uint32_t id_mask, SelectScratch& sc) {
    const uint32_t tid = threadIdx.x;
    const uint32_t cb = code_bits_of(id_mask);
    const bool coded = cb != 0u;
    const uint32_t code_shift = coded ? 32u - cb : 0u;
    // codes strictly between those of the bounds are inside for sure, codes beyond them outside for sure
    const int c_lo = lo == 0ull ? -1 : (int)depth_code((uint32_t)(lo >> 32), cb);
    const int c_hi = thr == ~0ull ? (1 << 16) : (int)depth_code((uint32_t)(thr >> 32), cb);
    __syncthreads();
    if (tid == 0) sc.count = 0;
    __syncthreads();
    constexpr uint32_t U = 8;  // entries per thread and trip: their loads, then their depth gathers, are all in flight together
    for (uint32_t base = 0; base < n; base += 256 * U) {
        uint32_t id[U];
        uint32_t inside = 0;  // bit u: entry u belongs to the front
        uint64_t key[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = base + u * 256 + tid;
            uint32_t e = 0u;
            bool valid = false;
            if (i < n) { e = bag[i]; valid = true; }
            id[u] = e & id_mask;
            const int c = (int)(e >> code_shift);
            const bool sure = valid && coded && c > c_lo && c < c_hi;
            const bool maybe = valid && (!coded || c == c_lo || c == c_hi);
            if (sure) inside |= 1u << u;
            key[u] = maybe ? depth_key(depths, id[u]) : 0ull;  // 0 <= lo: never inside
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            if (base + u * 256 >= n) break;  // workgroup-uniform
            const bool in = ((inside >> u) & 1u) || (key[u] > lo && key[u] <= thr);
            const uint64_t m = __ballot(in);
            if (m != 0ull) {  // wave-aggregated append
                const int leader = __builtin_ctzll(m);
                uint32_t wbase = 0;
                if ((int)(tid & 63) == leader) wbase = atomicAdd(&sc.count, (uint32_t)__builtin_popcountll(m));
                wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, leader);
                const uint32_t pos = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (in && pos < cap) sc.ids[pos] = id[u];
            }
        }
    }
    __syncthreads();
    return sc.count;
}

// Extracts the next front of a tile's bag: about `target` (at most cap <= FRONT_CAP, at least 1) of the ids whose key is above
// lo, depth-nearest first, into sc.ids (unsorted); `remaining` = how many ids above lo the bag holds.  The threshold is a
// quantile of up to 256 evenly spaced samples; should the front come out larger than cap (the count is tightly concentrated
// around the target, so this is the pathological case) the quantile is halved, down to the single nearest id, which always
// terminates.  Returns the front's length.
__device__ __forceinline__ uint32_t extract_front(const uint32_t* __restrict__ bag, uint32_t n, const float* __restrict__ depths, uint64_t lo,
                                                  uint32_t remaining, uint32_t target, uint32_t cap, uint32_t id_mask, SelectScratch& sc) {
    const uint32_t tid = threadIdx.x;
    if (remaining <= cap) return extract_pass(bag, n, depths, lo, ~0ull, cap, id_mask, sc);  // all that is left
    // sample: 256 evenly spaced entries, those above lo ranked among themselves
    const uint64_t mine = depth_key(depths, bag[(uint32_t)(((uint64_t)tid * n) >> 8)] & id_mask);
    const bool valid = mine > lo;
    __syncthreads();
    sc.sample[tid] = valid ? mine : ~0ull;
    if (tid == 0) sc.nvalid = 0;
    __syncthreads();
    if (valid) atomicAdd(&sc.nvalid, 1u);
    if (tid < 64) {  // wave 0 sorts the 256 sample keys in registers (4 per lane; every stage stays inside the wave: no barrier)
        SortCtx<4> c;
        c.t = tid;
        c.gidx = 4 * tid;
        c.xchg = nullptr;
#pragma unroll
        for (int r = 0; r < 4; r++) c.key[r] = sc.sample[4 * tid + r];
        sort_levels<4, 256>(c, 256u);
#pragma unroll
        for (int r = 0; r < 4; r++) sc.sorted[4 * tid + r] = c.key[r];  // samples at or below lo were set to ~0: they sort last
    }
    __syncthreads();
    const uint32_t k = sc.nvalid;
    for (;;) {
        uint64_t thr;
        if (k > 0 && target > 0) {
            thr = sc.sorted[min(k - 1, (uint32_t)(((uint64_t)k * target) / remaining))];
        } else {  // no usable sample, or even the nearest sample overshoots: the single nearest id above lo
            uint64_t best = ~0ull;
            for (uint32_t i = tid; i < n; i += 256) {
                const uint64_t key = depth_key(depths, bag[i] & id_mask);
                if (key > lo) best = min(best, key);
            }
            __syncthreads();
            sc.sample[tid] = best;
            __syncthreads();
            for (uint32_t j = 0; j < 256; j++) best = min(best, sc.sample[j]);
            thr = best;
        }
        const uint32_t F = extract_pass(bag, n, depths, lo, thr, cap, id_mask, sc);
        if (F <= cap) return F;
        target >>= 1;
    }
}

}  // namespace wg
