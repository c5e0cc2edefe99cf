// Internal shared declarations of the MI355X rasterizer library (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <type_traits>
#include "../../include/wg_rasterizer.h"

namespace wg {

constexpr int TILE_X = WG_TILE_X;
constexpr int TILE_Y = WG_TILE_Y;
constexpr size_t ALIGN = 256;

// ---- scratch carving (the role of obtain()/fromChunk(), rasterizer_impl.h:22-28, .cu:155-194) ----
template <typename T>
__host__ inline void carve(char*& chunk, T*& ptr, size_t count) {
    uintptr_t p = (reinterpret_cast<uintptr_t>(chunk) + ALIGN - 1) & ~(uintptr_t)(ALIGN - 1);
    ptr = reinterpret_cast<T*>(p);
    chunk = reinterpret_cast<char*>(ptr + count);
}

struct GeometryState {
    float* depths;
    int* radii;
    float4* splats;  // 3 float4 per Gaussian
    float* cov3D;
    unsigned char* clamped;
    ushort4* rects;  // tile rectangle (min.x, min.y, max.x, max.y)
    uint32_t* tiles_touched;
    uint32_t* point_offsets;
    uint16_t* band_list;  // large P only (g_band_list_min_p): [BIN_CHUNKS][8][chunk size] chunk-local indices of the Gaussians touching
    uint32_t* band_cnt;   // each XCD band of tiles, and their counts [BIN_CHUNKS][8]; the candidates of both scatter kernels
    float* grad_rec;      // backward only: one 48-byte gradient record per Gaussian (GRAD_REC_*), cleared at the start of every backward;
                          // grad_rec + 12 P: P more floats, the two-colour walk's thirteenth sum (render_bwd.hip: DUAL)
    char* scan_temp;
    size_t scan_temp_bytes;
    // band_lists: whether the two band-list arrays exist.  They are carved LAST, so that everything the backward pass and the
    // views address has the same offset either way (the decision is the forward call's alone, taken from its option snapshot).
    static GeometryState fromChunk(char*& chunk, size_t P, bool band_lists);
    static bool band_lists_possible(size_t P);  // chunk-local indices are 16-bit
};

// Gradient record of the per-tile backward pass (render_bwd.hip, RECORD): 12 floats = 48 bytes per Gaussian (three float4; a
// record lies within one 128-byte L2 line in 6 of 8 cases, in two otherwise), holding the raw wave-reduced sums
//   [0..2] sum(w dL_c)  [3] sum(q u)  [4] sum(q v)  [5] sum(|q| (0.5W|u| + 0.5H|v|))  [6..8] sum(q dx dx), sum(q dx dy), sum(q dy dy)
//   [9] sum(q)  [10..11] unused -- two-colour walk (DUAL): sum(w dL2_r), sum(w dL2_g); sum(w dL2_b) goes to grad_aux[id].
//   preprocess_bwd.hip turns them into the reference's four arrays (+ dL_dcolor2).
constexpr int GRAD_REC_FLOATS = 12;

struct BinStats {  // read back by the host once per forward (the reference's num_rendered sync point)
    uint32_t num_rendered;
    uint32_t max_tile_count;
    uint32_t split_active;  // 1: the near / far split is on for this frame (SplitState::near_code is a real threshold)
    uint32_t spec_fail;     // speculative forward (api.hip): 1 = this frame does not fit what the host enqueued ahead of the count (more
                            // instances than the binning buffer holds, or a list longer than the launched sort network covers):
                            // every guarded kernel behind the scan returns at once and the host re-issues the tail with the real sizes
};
// What a speculative forward pass promises the kernels it enqueues before the instance count is known (all zero: no speculation).
struct SpecLimits {
    uint32_t capacity = 0;   // instances the binning buffer holds
    uint32_t max_list = 0;   // longest per-tile list the launched sort path covers (0xffffffff: any, i.e. the lazy front sort)
};

// Near / far split of dense frames (binning.hip): a frame-wide depth-code threshold, chosen on the device from a histogram of the
// visible Gaussians' 12-bit depth codes weighted by their tile counts, such that about `near_per_tile` instances per tile are
// "near".  Only the near instances are scattered and front-sorted; a tile whose pixels are still accumulating when its near
// instances are used up gets its far ones scattered afterwards (rare).  near_code == SPLIT_OFF: no split (every instance is near).
constexpr uint32_t SPLIT_BITS = 12, SPLIT_BINS = 1u << SPLIT_BITS, SPLIT_OFF = 0xffffffffu;
// instances per tile from which a frame counts as dense for the split (measured on the bench scene with scaled Gaussians: forcing
// the split loses 2.5 % at 911 per tile and gains 5 / 7 / 10 % at 1180 / 1400 / 1640)
constexpr uint32_t SPLIT_DENSE_AVG = 1100;
// The per-Gaussian kernels' once-per-frame streams -- the SH block in, dL_dsh out -- as NON-TEMPORAL accesses (`nt`: no allocation in the caches on
// the way): at 1 M Gaussians the 192 MB SH block otherwise flushes the records, lists and image state the other kernels re-read out of the L2s /
// the memory-side cache (preprocess 0.0706 -> 0.0581 ms, preprocess_backward 0.1023 -> 0.0960, the render kernels ~1 % each); at 10 M nothing
// fits anyway and the hint costs ~2 % (EXPERIMENTS.md R6.11).  A template parameter of the kernels (SH mode 2), chosen per launch (option "sh_stream").
typedef float wg_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 stream_load4(const float4* p) {
    if constexpr (NT) {
        const wg_v4f v = __builtin_nontemporal_load(reinterpret_cast<const wg_v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void stream_store4(float4* p, float4 v) {
    if constexpr (NT) {
        wg_v4f t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<wg_v4f*>(p));
    } else {
        *p = v;
    }
}

struct SplitState {
    uint32_t near_code;  // instances of Gaussians with depth_code(depth, SPLIT_BITS) <= near_code are near
    uint32_t need_far;   // bit b: set by the first fix-up phase when a tile of XCD band b ran out of near instances
    uint32_t far_tiles;  // how many tiles did
    uint32_t aim;        // the aimed near instances per tile this frame's threshold was picked for (goes back to the host with the far-phase report)
};

// Host-visible (pinned, mapped, coherent) mailbox the tile scan writes the same two numbers to, followed by a sequence
// number with system-scope release: the host polls it instead of paying a D2H copy + stream synchronise.
struct HostMailbox {
    // Two self-describing 64-bit words, each stored with ONE relaxed system-scope store (an aligned 8-byte store reaches pinned
    // host memory whole): the frame's sequence number in the upper half of both, so the host takes a pair only when both carry the
    // number it waits for -- no release fence on the device, which at system scope writes the whole L2 back first (16 MB of
    // chunk histograms at the headline scene).
    //   word0 = seq << 32 | num_rendered
    //   word1 = seq << 32 | spec_fail << 31 | split_active << 30 | min(max_tile_count, 2^30 - 1)
    unsigned long long word0, word1;
    // written at the END of a split frame (fix-up phase 1) with ONE 8-byte store: low word = 1 + the number of tiles that needed far instances (24 bits)
    // + the number of XCD bands with such a tile << 24 (the far scatter walks the far Gaussians of the flagged bands: its cost goes with the
    // bands); high word = the near aim that frame ran with (SplitState::aim).  0 = nothing new.  Read by the host at the start of a LATER frame
    // as a hint only (no waiting: it may be a frame or two old -- hence the aim inside the report, not the host's current one)
    unsigned long long far_report;
    uint32_t pad[2];
};
constexpr uint32_t MAILBOX_MAX_LIST = (1u << 30) - 1u;

struct ImageState {
    float* final_T;  // MUST stay first: documented in wg_rasterizer.h
    float* accum;    // MUST stay second: 1 - final_T, written by the forward kernels when a tile completes (the binding returns a
                     // view of it as `accumulation` instead of launching an elementwise kernel per call)
    uint32_t* n_contrib;
    uint2* ranges;
    uint32_t* tile_last;
    uint32_t* tile_count;   // per-tile instance count
    uint32_t* tile_offset;  // exclusive prefix sum of tile_count
    uint32_t* chunk_hist;   // [chunks][tiles] per-chunk tile histogram, turned into per-chunk bases by the column scan
    uint32_t* order_bwd;    // launch order of the backward render: per XCD band, tiles by descending walked length
    uint32_t* order_fwd;    // launch order of the forward render (binning.hip: forward_order_kernel), when the per-camera history is on
    uint32_t* order_key;    // [4] {the camera's row in the library's order table, tag lo, tag hi, 1 = the row held this camera}
    uint32_t* seg_end;      // lazy sort: length of the tile's sorted front after the first round (== count when sorted in full)
    uint32_t* tile_state;   // lazy sort: 0xffffffff = tile finished, else the list length the forward pass has consumed
    BinStats* stats;
    uint32_t* tile_near;    // near / far split: per-tile count of near instances (== tile_count without a split)
    uint32_t* far_cursor;   //                   per-tile append cursor of the far scatter
    uint32_t* code_hist;    //                   [SPLIT_BINS] tile-count-weighted histogram of the visible Gaussians' depth codes
    SplitState* split;
    uint32_t* scan_ticket;  // fused column + tile scan: workgroups done so far
    static ImageState fromChunk(char*& chunk, size_t N, size_t tiles);
};

// Binning scratch.  Two layouts share the point_list prefix (all the backward pass needs):
//   tile-sort path (default): point_list u32[R] | bucket_ids u32[R]
//   global-sort fallback    : point_list u32[R] | point_list_unsorted u32[R] | keys u64[R] | keys_unsorted u64[R] | temp
struct BinningState {
    uint32_t* point_list;
    uint32_t* bucket_ids;   // Gaussian ids grouped by tile, unsorted within a tile
    uint32_t* point_list_unsorted;
    uint64_t* keys;
    uint64_t* keys_unsorted;
    char* sort_temp;
    size_t sort_temp_bytes;
    static BinningState fromChunk(char*& chunk, size_t R, bool global_sort);
};

constexpr uint32_t TILE_SORT_MAX = 8192;  // longest per-tile list the register sort handles (32 keys per thread)
constexpr int BIN_CHUNKS = 512;           // Gaussian chunks (= workgroups) of the LDS counting sort
constexpr int BIN_MAX_TILES = 36864;      // tiles*4 B must fit one workgroup's LDS (144 KiB): up to 4K frames

size_t query_scan_temp_bytes(size_t P);
size_t query_sort_temp_bytes(size_t R);

// Optional per-Gaussian affine on the SH coefficients before their evaluation (include/wg_rasterizer.h: wg_sh_tone):
//   x = min(sh[k][c], pre_clamp);  t = x * mul[c] + (k == 0 ? offset[c] : 0);  used value = min(t, post_clamp)
// evaluated with separate multiply and add (what the PyTorch elementwise chain it replaces rounds to).  enabled == 0: untouched.
struct ShTone {
    int enabled = 0;
    const float* mul = nullptr;     // [P,3] or null (= 1)
    const float* offset = nullptr;  // [P,3] or null (= 0)
    float pre_clamp = 0.f, post_clamp = 0.f;
    float* dL_dmul = nullptr;       // backward: [P,3] each, written for every Gaussian (zeros when culled)
    float* dL_doffset = nullptr;
    // second != 0 (wg_forward_args::sh_second): a SECOND colour set from the same coefficients through a tone of its own, composited in
    // the same walk (render_fwd.hip / render_bwd.hip: DUAL); its colour-clamp flags take bits 3-5 of GeometryState::clamped
    int second = 0;
    const float* mul2 = nullptr;
    const float* offset2 = nullptr;
    float pre_clamp2 = 0.f, post_clamp2 = 0.f;
    float* dL_dmul2 = nullptr;
    float* dL_doffset2 = nullptr;
};

// the used value and, for the backward pass, what it was made of
__device__ __forceinline__ float tone_value(float raw, float m, float o, float pre, float post, float& xin, float& t) {
    xin = fminf(raw, pre);
    t = __fadd_rn(__fmul_rn(xin, m), o);
    return fminf(t, post);
}

// kernel argument of the TONE instantiations only: the plain kernels keep the argument block (and the code) they had
struct NoTone {};
template <bool TONE>
using ToneArg = typename std::conditional<TONE, ShTone, NoTone>::type;

struct FwdParams {
    int P, D, M, W, H, gx, gy;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* colors_precomp2;   // second colour set (two-colour walk) or null: written to the record's spare floats r1.z, r2.z, r2.w
    const float* opacities;
    const float* scales;
    float scale_modifier;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* cam_pos;
    float tan_fovx, tan_fovy, focal_x, focal_y, kernel_size;
    int prefiltered;
    const float* filter_3D = nullptr;   // raw-parameter mode (wg_raw_gaussians): opacities / scales / rotations above are the caller's RAW
                                        // parameters and get_gaussians() (method.py:1060-1086) runs inside the kernel (wg_act.h)
    bool nt_stream = false;             // the SH block as non-temporal loads (stream_load4 above)
};

// kernels / stages (each launches on `stream`, returns hipGetLastError())
hipError_t launch_preprocess(const FwdParams& p, const ShTone& tone, const GeometryState& g, int* radii_out, bool geom_only, hipStream_t stream);
// the colour half of a split frame (preprocess.hip: GEOM_ONLY): far = false the near Gaussians, far = true the others if some tile asked for them
struct SplitState;
hipError_t launch_sh_colour(const FwdParams& p, const GeometryState& g, const SplitState* split, bool far, hipStream_t stream);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, unsigned char* present, hipStream_t stream);
hipError_t launch_recolor(int P, const GeometryState& src, const GeometryState& dst, const float* colors, int* radii_out, hipStream_t stream);
hipError_t run_scan(const GeometryState& g, int P, hipStream_t stream);
hipError_t launch_scan_overflow_check(const GeometryState& g, int P, uint32_t* flag, hipStream_t stream);  // *flag = 1: the 32-bit scan wrapped
hipError_t launch_duplicate_keys(int P, const GeometryState& g, const BinningState& b, int gx, hipStream_t stream);
hipError_t run_sort(const BinningState& b, int R, int end_bit, hipStream_t stream);
hipError_t launch_tile_ranges(int R, const BinningState& b, const ImageState& img, int tiles, hipStream_t stream);
// clear != nullptr: the same launch zeroes clear_floats floats (the backward pass's gradient records)
// period: 0 = descending cost; > 0 = dealt in rounds of that many tiles, every other round reversed (binning.hip: order_band)
hipError_t launch_tile_order(const uint32_t* cost_or_null, const uint2* ranges_or_null, uint32_t* order, int tiles, float* clear, size_t clear_floats,
                             int period, hipStream_t stream);
// the forward render kernel's launch order from the per-camera history table (table: 2 * slots tag words, then slots rows of `stride` costs;
// nullptr = no table: nothing is ordered); order / key_out: the frame's ImageState::order_fwd / order_key
struct FwdOrderArgs {
    const float* viewmatrix = nullptr;
    const float* projmatrix = nullptr;
    int W = 0, H = 0;
    uint32_t* table = nullptr;
    uint32_t slots = 0, stride = 0;
    uint32_t* order = nullptr;
    uint32_t* key_out = nullptr;
    uint32_t period = 0;
};
hipError_t launch_forward_order(const FwdOrderArgs& a, int tiles, hipStream_t stream);   // stand-alone; launch_tile_scan carries it along otherwise
// split == nullptr: no near / far split attempted (small P): the kernels are exactly the ones without it
hipError_t launch_split_threshold(int P, const GeometryState& g, const ImageState& img, int tiles, bool force, uint32_t near_per_tile,
                                  hipStream_t stream);
// box: count through a difference grid + two prefix passes (four atomics per Gaussian) instead of one atomic per instance
hipError_t launch_tile_count(int P, const GeometryState& g, const ImageState& img, int gx, int tiles, bool split, bool box, bool fused_scan,
                             hipStream_t stream);
hipError_t launch_tile_scan(const ImageState& img, int tiles, HostMailbox* mailbox_dev, uint32_t seq, bool split, const SpecLimits& spec,
                            bool fused_scan, const FwdOrderArgs& fo, hipStream_t stream);   // fo.table != nullptr (and !fused_scan): eight more workgroups order the bands
// guard (everywhere below): nullptr, or the frame's BinStats -- the kernel returns at once when spec_fail is set there
hipError_t launch_tile_scatter_far(int P, const GeometryState& g, const ImageState& img, const BinningState& b, int gx, int tiles, int code_bits,
                                   const BinStats* guard, hipStream_t stream);
// code_bits > 0: bucket entries carry a coarse depth code of that width above the id (wg_sort.h: depth_code); only the lazy
// sort reads it
hipError_t launch_tile_scatter(int P, const GeometryState& g, const ImageState& img, const BinningState& b, int gx, int tiles,
                               uint32_t num_rendered, int code_bits, int staged_scatter, int staged_cap, bool split, const BinStats* guard,
                               hipStream_t stream);
struct LazyConfig {
    bool enabled = true;
    uint32_t min_len = 1024;  // tiles listing more than this are front-split instead of sorted in full (the lazy path as a whole is
                              // taken when the longest list exceeds 5/4 of it: below, a few lists on the 2048-key network cost less
                              // than coded bucket entries and the fix-up launch)
    uint32_t target = 820;    // aimed front length of the first round (the 1024-key network)
    uint32_t cap = 2048;      // hard bound of a front (the 2048-key network)
};
// Every option wg_set_option() can change (include/wg_rasterizer.h).  The library keeps one instance behind a mutex; each C-ABI call
// takes ONE copy at its start and hands it down, so a call sees a consistent set whatever other host threads set meanwhile.
struct Options {
    LazyConfig lazy;
    int staged_scatter = -1;          // -1 auto / 0 / 1
    int staged_cap = 0;               // staging-area entries, 0 = what the LDS budget allows (tests: multi-pass)
    int band_list_min_p = 2000000;    // from here on the scatter kernels read per-band candidate lists instead of whole chunks
    int depth_codes = 1;              // 0 / 1 / 8..12: off (as for P > 2^24) / automatic width / forced width (tests)
    int speculative = 1;              // speculative forward (api.hip): enqueue everything behind the instance count before it is known
    int spec_margin_pct = 25;         //   binning buffer = the recent frames' largest count + this margin
    int fused_scan = 0;               // column scan + tile scan in one launch (last-workgroup hand-over); see EXPERIMENTS.md for the A/B
    int geometry_reuse = 0;           // read by the torch binding (_C.py; opt-in since round 4): consecutive calls over identical geometry and
                                      // camera share the projection and the binning of the first (wg_forward_args::recolor)
    int exact_compositing = 1;        // 1: the render kernels take every skip / stop decision (power > 0, alpha < 1/255, T (1 - alpha) < 1e-4) on values
                                      // computed with the reference's own float32 operations (wg_alpha.h): n_contrib, final_T and the blended set are
                                      // the reference's bit for bit.  0: exp2 of a pre-scaled fused form (round 1-3; ~2 ppm of pixels flip).  Must not
                                      // change between a frame's forward and backward call.
    int grad_record = 1;              // 0: the per-tile backward accumulates into the four arrays themselves (A/B)
    int deterministic_backward = 0;   // 1: per-instance slots + an ordered per-Gaussian sum instead of float atomics (bit-reproducible)
    int near_split = -1;              // near / far split of dense frames: -1 automatic (P >= band_list_min_p or a dense previous frame, and >= SPLIT_DENSE_AVG =
                                      // 1100 instances per tile) / 0 off / 1 whenever possible (tests)
    int near_per_tile = 0;            // aimed near instances per tile; 0 = adapted from 1.1 x lazy.target down (near_adapt) or that value itself
    int near_adapt = 1;               // 1: the aim follows the frames' far-phase reports per host thread (api.hip: NearAdapt); 0: fixed
    int box_count = -1;               // tile counting through a difference grid: -1 automatic (with the split's "large or dense" rule) / 0 / 1
    bool force_global_sort = false;   // exercise the fallback binning path
    bool use_mailbox = true;          // 0 restores the copy + synchronise read-back
    int forward_order = 1;            // launch order of the forward render kernel from the per-camera tile-cost history (0 = image order, no table)
    int forward_order_slots = 2048;   //   rows of that table (x 9216 tiles x 4 B = 75 MB of device memory, allocated at the first forward call)
    int order_period = 128;           //   rounds of the snake dealing (the hardware's placement period per XCD; 0 = plain descending order)
    int backward_order_period = 0;    //   the same for the backward kernel's order (0: descending, its tail is dealt dynamically)
    int lazy_colour = 1;              // frames that attempt the near / far split with plain SH colours: colour the near Gaussians only, the far ones when asked for
    int lazy_colour_min_p = 4000000;  //   from this many Gaussians on (10 M / 4K: 583 -> 647 fps; 3 M / 1080p fwd+bwd: 719 -> 711 iter/s, 1 M dense: -0.4 %)
    int sh_stream = -1;               // the SH block in / dL_dsh out as non-temporal accesses: 1 on, 0 off, -1 up to sh_stream_max_p Gaussians
    int sh_stream_max_p = 6000000;    //   (wg_common.h: stream_load4; a gain while the rest of the frame's state fits the caches, a small loss at 10 M)
};
hipError_t launch_tile_sort_lazy(const ImageState& img, const BinningState& b, const GeometryState& g, int tiles, int code_bits,
                                 const LazyConfig& lazy, bool split, const BinStats* guard, hipStream_t stream);
// phase 0: the near bag (all of the bucket without a split); phase 1: the far bag of the tiles that asked for it
hipError_t launch_render_fixup(int code_bits, int W, int H, int gx, int gy, const ImageState& img, const BinningState& b, const GeometryState& g,
                               const float* subpixel_offset, const float* background, float* out_color, float* out_color2, const LazyConfig& lazy,
                               bool split, int phase, bool exact, HostMailbox* mailbox_dev, const BinStats* guard, hipStream_t stream);
hipError_t launch_tile_sort(const ImageState& img, const BinningState& b, const GeometryState& g, int tiles, uint32_t max_count,
                            const BinStats* guard, hipStream_t stream);
// order_table != nullptr: img.order_fwd / img.order_key are valid (launch_forward_order ran): tiles launch in that order and write their
// walked lengths back into the camera's row of the table
hipError_t launch_render_forward(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                 const GeometryState& g, const float* subpixel_offset, const float* background,
                                 float* out_color, float* out_color2, bool lazy, bool exact, const BinStats* guard, uint32_t* order_table,
                                 uint32_t order_slots, uint32_t order_stride, hipStream_t stream);
// the compositing of a frame whose binning and per-pixel stops are known (img.tile_last, img.n_contrib of an earlier pass over the
// same geometry): each tile walks exactly its list's first tile_last entries and stores final outputs
// capturable forward (api.hip: wg_forward_args::binning_capacity): when the frame did not fit (BinStats::spec_fail) the image and the
// accumulation become NaN and tile_last / n_contrib zero
hipError_t launch_poison_unfit(const ImageState& img, int W, int H, int tiles, float* out_color, float* out_color2, hipStream_t stream);
hipError_t launch_render_forward_replay(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                        const GeometryState& g, const float* subpixel_offset, const float* background,
                                        float* out_color, bool exact, hipStream_t stream);
hipError_t launch_render_backward(int W, int H, int gx, int gy, const ImageState& img, const BinningState& b,
                                  const GeometryState& g, const float* subpixel_offset, const float* background,
                                  const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                  float* dL_dcolor, bool record, bool exact, const float* dL_dpix2, float* det_slots, unsigned char* det_flags, size_t slot_capacity, int P,
                                  hipStream_t stream);  // det_slots != nullptr: deterministic mode (det_flags: one byte per slot, cleared)

struct BwdParams {
    int P, D, M, W, H;
    const float* means3D;
    const float* shs;
    const float* scales;
    float scale_modifier;
    const float* rotations;
    const float* cov3D;  // precomputed or geometry-state copy
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    float tan_fovx, tan_fovy, focal_x, focal_y, kernel_size;
    const int* radii;
    float* dL_dcolor2 = nullptr;    // two-colour walk: [P,3], written from the record's floats 10, 11 and grad_aux
    const float* filter_3D = nullptr;      // raw-parameter mode: scales / rotations above are RAW, raw_opacities the raw opacities; dL_dscale /
    const float* raw_opacities = nullptr;  // dL_drot / dL_dopacity come out as the gradients of the RAW parameters
    bool nt_stream = false;                // the SH block in and dL_dsh out as non-temporal accesses (stream_load4 / stream_store4)
};
// record: the four arrays are OUTPUTS computed from g.grad_rec (see GRAD_REC_*); otherwise inputs accumulated by the per-tile pass
hipError_t launch_preprocess_backward(const BwdParams& p, const ShTone& tone, const GeometryState& g, float* dL_dmean2D,
                                      float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                      float* dL_drot, bool record, hipStream_t stream);

uint32_t higher_msb(uint32_t n);

}  // namespace wg
