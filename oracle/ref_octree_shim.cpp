// ref_octree_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" wrapper around the REFERENCE's own k-NN
// engine, compiled from the sources where they lie:
//   /root/reference/super_odometry/include/super_odometry/flann/octree.h   (nanoflann::Octree)
//   /root/reference/super_odometry/include/super_odometry/flann/nanoflann.h (KNNResult)
// Nothing from those files is copied into this repository; oracle/Makefile passes their directory
// with -I and writes the result to oracle/_ref/libref_octree.so (git-ignored, travels with gpurun).
// Calls mirror LocalMap.h:638 (initialize) and LocalMap.h:516-520 (knnNeighbors<L2Distance>).
#include <cstddef>
#include <cstdint>
#include <vector>

#include "octree.h"

namespace {
struct Pt { float x, y, z; };  // any struct with public x,y,z (octree.h:23-62 access traits)
struct Handle {
  std::vector<Pt> pts;  // the octree keeps a raw pointer to this container (octree.h:363)
  nanoflann::Octree<Pt, std::vector<Pt>> tree;
};
}  // namespace

namespace {
// Oracle-B: one tree per LocalMap block, like MapBlock::octree_surf_ (LocalMap.h:45-53), rebuilt when the block's point
// array changed (LocalMap.h:638 rebuilds after every insert into the block)
struct CubeTree { const float* xyz = nullptr; size_t n = 0; Handle* h = nullptr; };
CubeTree g_cube_trees[21 * 21 * 11];
}  // namespace

extern "C" {
// signature = orc_knn_hook_t (oracle/so_oracle.h); single-threaded by design (the reference's correspondence loop is serial)
void ref_octree_cube_knn(int cube_ind, const float* xyz, size_t n, const float q[3], int k, int64_t* idx, float* d2) {
  CubeTree& t = g_cube_trees[cube_ind];
  if (t.xyz != xyz || t.n != n || !t.h) {
    delete t.h;
    t.h = new Handle();
    t.h->pts.resize(n);
    for (size_t i = 0; i < n; ++i) t.h->pts[i] = Pt{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    t.h->tree.initialize(t.h->pts);
    t.xyz = xyz; t.n = n;
  }
  std::vector<size_t> k_indices(k, 0);   // LocalMap.h:516: std::vector<size_t> k_indices(k)
  std::vector<float> k_d2(k, 0.f);
  Pt qq{q[0], q[1], q[2]};
  t.h->tree.knnNeighbors<nanoflann::L2Distance<Pt>>(qq, (size_t)k, k_indices.data(), k_d2.data());
  for (int j = 0; j < k; ++j) { idx[j] = (int64_t)k_indices[j]; d2[j] = k_d2[j]; }
}
void ref_octree_cube_reset(void) {
  for (CubeTree& t : g_cube_trees) { delete t.h; t = CubeTree(); }
}
void* ref_octree_build(const float* xyz, size_t n) {
  Handle* h = new Handle();
  h->pts.resize(n);
  for (size_t i = 0; i < n; ++i) h->pts[i] = Pt{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  h->tree.initialize(h->pts);  // default OctreeParams: bucket 32, copyPoints=false
  return h;
}
void ref_octree_free(void* hp) { delete static_cast<Handle*>(hp); }
// Same buffer protocol as LocalMap.h:516-520: k_indices value-initialised to 0, distances resized.
void ref_octree_knn(void* hp, const float* q_xyz, size_t nq, int k, int64_t* idx_out, float* d2_out) {
  Handle* h = static_cast<Handle*>(hp);
  std::vector<size_t> k_indices(k);
  std::vector<float> k_d2(k);
  for (size_t i = 0; i < nq; ++i) {
    std::fill(k_indices.begin(), k_indices.end(), 0);
    std::fill(k_d2.begin(), k_d2.end(), 0.f);
    Pt q{q_xyz[3 * i], q_xyz[3 * i + 1], q_xyz[3 * i + 2]};
    h->tree.knnNeighbors<nanoflann::L2Distance<Pt>>(q, (size_t)k, k_indices.data(), k_d2.data());
    for (int j = 0; j < k; ++j) {
      idx_out[i * k + j] = (int64_t)k_indices[j];
      d2_out[i * k + j] = k_d2[j];
    }
  }
}
}
