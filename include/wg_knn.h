/*
 * wg_knn.h -- C-ABI of the MI355X replacement for simple_knn._C.distCUDA2.
 *
 * Replaces SimpleKNN::knn (submodules/simple-knn/simple_knn.h:15-19, simple_knn.cu:185-221), which the reference's
 * torch binding calls from distCUDA2 (spatial.cu:15-26; used once, at initialisation, wildgaussians/method.py:1001).
 * Result: mean_dists[i] = mean of the squared distances from point i to its 3 nearest other points (float32).
 * All pointers are device pointers; points is float[P*3], mean_dists float[P].  The caller supplies the scratch
 * (wg_knn_scratch_size bytes, 256-byte aligned) -- the reference cudaMalloc'ed and used thrust vectors internally.
 * Returns 0 or a negative wg_status (wg_rasterizer.h).  No host synchronisation.
 */
#ifndef WG_KNN_H_INCLUDED
#define WG_KNN_H_INCLUDED
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t wg_knn_scratch_size(int P);
int wg_knn_mean_dist2(int P, const float* points, float* mean_dists, char* scratch, size_t scratch_bytes, void* stream);
#ifdef __cplusplus
}
#endif
#endif
