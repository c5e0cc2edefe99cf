/*
 * wg_knn_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for SURVEY.md 8f row N1: simple_knn._C.distCUDA2.
 *
 * Literal restatement of submodules/simple-knn/simple_knn.cu (SimpleKNN::knn, :185-221 and its kernels :45-183) and
 * spatial.cu:15-26: mean squared distance of every point to its 3 nearest neighbours, found through a Morton-ordered
 * list cut into boxes of 1024 points.  Quirks kept: the bounding box reduction starts from (0,0,0) (simple_knn.cu:191,
 * so the box always contains the origin); float -> uint conversion of the normalised coordinate truncates; the
 * 3-best list is updated by insertion (:129-143); a point's own slot is skipped by sorted position, so exact duplicates
 * count as neighbours at distance 0.
 *
 * wgo_knn_bruteforce() is an independent O(P^2) statement of the same quantity used to cross-check the restatement.
 * PARITY STATUS: PINNED against outputs of the reference itself -- simple_knn.cu compiles for gfx950 with hipcc where it
 * lies (oracle/ref_hip/Makefile); run on an MI355X it produced tests/golden/ref_hip_knn_golden.npz, and tests/test_knn.py
 * holds this restatement to it bit for bit (-ffp-contract=off build of the reference; <= 1e-6 relative against its
 * default-contraction build, which differs from the other by an ulp in 5-10 % of the points).
 * Build: -ffp-contract=off (see oracle/Makefile).  Only tests/ may load this library.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WGO_API __attribute__((visibility("default")))
#define BOX_SIZE 1024

typedef struct { float x, y, z; } f3;
typedef struct { f3 minn, maxx; } MinMax;

/* simple_knn.cu:45-52 */
static uint32_t prepMorton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
static uint32_t f2u(float v) { return (v != v || v <= 0.0f) ? 0u : (v >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)v); } /* device cvt semantics */
/* simple_knn.cu:54-61 */
static uint32_t coord2Morton(f3 c, f3 minn, f3 maxx) {
    uint32_t x = prepMorton(f2u(((c.x - minn.x) / (maxx.x - minn.x)) * (float)((1 << 10) - 1)));
    uint32_t y = prepMorton(f2u(((c.y - minn.y) / (maxx.y - minn.y)) * (float)((1 << 10) - 1)));
    uint32_t z = prepMorton(f2u(((c.z - minn.z) / (maxx.z - minn.z)) * (float)((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}
/* simple_knn.cu:117-127 */
static float distBoxPoint(const MinMax* box, f3 p) {
    f3 diff = {0, 0, 0};
    if (p.x < box->minn.x || p.x > box->maxx.x) diff.x = fminf(fabsf(p.x - box->minn.x), fabsf(p.x - box->maxx.x));
    if (p.y < box->minn.y || p.y > box->maxx.y) diff.y = fminf(fabsf(p.y - box->minn.y), fabsf(p.y - box->maxx.y));
    if (p.z < box->minn.z || p.z > box->maxx.z) diff.z = fminf(fabsf(p.z - box->minn.z), fabsf(p.z - box->maxx.z));
    return diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
}
/* simple_knn.cu:129-143, K = 3 */
static void updateKBest(f3 ref, f3 point, float* knn) {
    f3 d = {point.x - ref.x, point.y - ref.y, point.z - ref.z};
    float dist = d.x * d.x + d.y * d.y + d.z * d.z;
    for (int j = 0; j < 3; j++)
        if (knn[j] > dist) {
            float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
}

/* SimpleKNN::knn, simple_knn.cu:185-221; points [P,3], meanDists [P] */
WGO_API void wgo_knn(int P, const float* pts, float* meanDists) {
    if (P <= 0) return;
    const f3* points = (const f3*)pts;
    f3 minn = {0, 0, 0}, maxx = {0, 0, 0}; /* init = {0,0,0}, :191 */
    for (int i = 0; i < P; i++) {
        minn.x = fminf(minn.x, points[i].x); minn.y = fminf(minn.y, points[i].y); minn.z = fminf(minn.z, points[i].z);
        maxx.x = fmaxf(maxx.x, points[i].x); maxx.y = fmaxf(maxx.y, points[i].y); maxx.z = fmaxf(maxx.z, points[i].z);
    }
    uint32_t* morton = (uint32_t*)malloc((size_t)P * 4);
    uint32_t* idx_a = (uint32_t*)malloc((size_t)P * 4);
    uint32_t* idx_b = (uint32_t*)malloc((size_t)P * 4);
    uint32_t* key_b = (uint32_t*)malloc((size_t)P * 4);
    for (int i = 0; i < P; i++) {
        morton[i] = coord2Morton(points[i], minn, maxx);
        idx_a[i] = (uint32_t)i;
    }
    /* stable LSD radix sort of (morton, index), the contract of cub::DeviceRadixSort::SortPairs (:210-213) */
    uint32_t *ka = morton, *kb = key_b, *va = idx_a, *vb = idx_b;
    for (int shift = 0; shift < 32; shift += 8) {
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (int i = 0; i < P; i++) count[((ka[i] >> shift) & 255) + 1]++;
        for (int b = 0; b < 256; b++) count[b + 1] += count[b];
        for (int i = 0; i < P; i++) {
            size_t d = count[(ka[i] >> shift) & 255]++;
            kb[d] = ka[i];
            vb[d] = va[i];
        }
        uint32_t* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    const uint32_t* indices = va; /* sorted */
    const int num_boxes = (P + BOX_SIZE - 1) / BOX_SIZE;
    MinMax* boxes = (MinMax*)malloc((size_t)num_boxes * sizeof(MinMax));
    /* boxMinMax, :78-115 */
    for (int b = 0; b < num_boxes; b++) {
        MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
        for (int i = b * BOX_SIZE; i < P && i < (b + 1) * BOX_SIZE; i++) {
            f3 p = points[indices[i]];
            me.minn.x = fminf(me.minn.x, p.x); me.minn.y = fminf(me.minn.y, p.y); me.minn.z = fminf(me.minn.z, p.z);
            me.maxx.x = fmaxf(me.maxx.x, p.x); me.maxx.y = fmaxf(me.maxx.y, p.y); me.maxx.z = fmaxf(me.maxx.z, p.z);
        }
        boxes[b] = me;
    }
    /* boxMeanDist, :147-183 */
#pragma omp parallel for schedule(dynamic, 256)
    for (int idx = 0; idx < P; idx++) {
        f3 point = points[indices[idx]];
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int i = (idx - 3 > 0 ? idx - 3 : 0); i <= (P - 1 < idx + 3 ? P - 1 : idx + 3); i++) {
            if (i == idx) continue;
            updateKBest(point, points[indices[i]], best);
        }
        float reject = best[2];
        best[0] = best[1] = best[2] = FLT_MAX;
        for (int b = 0; b < num_boxes; b++) {
            float dist = distBoxPoint(&boxes[b], point);
            if (dist > reject || dist > best[2]) continue;
            for (int i = b * BOX_SIZE; i < (P < (b + 1) * BOX_SIZE ? P : (b + 1) * BOX_SIZE); i++) {
                if (i == idx) continue;
                updateKBest(point, points[indices[i]], best);
            }
        }
        meanDists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    free(morton); free(idx_a); free(idx_b); free(key_b); free(boxes);
}

/* independent O(P^2) statement: mean of the three smallest squared distances to the other points */
WGO_API void wgo_knn_bruteforce(int P, const float* pts, float* meanDists) {
    const f3* points = (const f3*)pts;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = 0; j < P; j++)
            if (j != i) updateKBest(points[i], points[j], best);
        meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

WGO_API uint32_t wgo_morton(float x, float y, float z, const float* minn, const float* maxx) {
    f3 c = {x, y, z}, a = {minn[0], minn[1], minn[2]}, b = {maxx[0], maxx[1], maxx[2]};
    return coord2Morton(c, a, b);
}
