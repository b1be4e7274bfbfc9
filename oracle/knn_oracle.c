/* CPU oracle of the K-nearest-neighbour mean squared distance -- TEST INFRASTRUCTURE ONLY: tests/ and
 * __graft_entry__.smoke() may use it as the checker; the product path never does.
 *
 * PARITY UNPINNED: the reference binds `dist3knn` / `dist10knn` / `meanDistFromReferencePcd` from its simple-knn fork
 * [REF /root/reference/scene/gaussian_model.py:16,151; inpainting_pipeline/2_condition_preparation/2_generate_inpainted_mask.py:27,71-73],
 * an un-vendored submodule (/root/reference/.gitmodules, no pinned SHA) with no tests or golden vectors in the reference.
 * Restated here is the published simple-knn definition it descends from (graphdeco-inria/simple-knn `distCUDA2`):
 * out[i] = (sum of the K smallest squared Euclidean distances from point i to the OTHER points) / K, float32,
 * distance = (dx*dx + dy*dy) + dz*dz.  Brute force, so it is exact by construction. */
#include <float.h>
#include <math.h>
#include <stdint.h>

/* query == reference (self = 1): point j == i is skipped.  K <= 16. */
void so_knn_mean_dist2(int nq, const float* query, int nr, const float* reference, int K, int self, int take_sqrt, float* out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < nq; ++i) {
        float best[16];
        for (int k = 0; k < K; ++k) best[k] = FLT_MAX;
        const float qx = query[3 * i], qy = query[3 * i + 1], qz = query[3 * i + 2];
        for (int j = 0; j < nr; ++j) {
            if (self && j == i) continue;
            const float dx = qx - reference[3 * j], dy = qy - reference[3 * j + 1], dz = qz - reference[3 * j + 2];
            float d = (dx * dx + dy * dy) + dz * dz;
            if (!(d < best[K - 1])) continue;
            for (int k = 0; k < K; ++k) {   /* ascending insertion */
                if (d < best[k]) { const float t = best[k]; best[k] = d; d = t; }
            }
        }
        float sum = best[0];
        for (int k = 1; k < K; ++k) sum += best[k];
        const float mean = sum / (float)K;
        out[i] = take_sqrt ? sqrtf(mean) : mean;
    }
}
