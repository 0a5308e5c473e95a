// local_map.cpp -- see local_map.h.  Host C++; no device code here.
#include "local_map.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>

#include "so_math.h"

namespace soicp {

static inline int cidx(int i, int j, int k) { return i + kMapW * j + kMapW * kMapH * k; }

int cells_per_cube(float plane_res, double* cell_size) {
  // gate radius: the reference rejects a match when d2[4] > 3*planeRes (float product, LidarSlam.cpp:526,741)
  const double r_max = std::sqrt((double)(3 * plane_res));
  int nc = (int)std::floor(kCube / (r_max * 1.005));
  if (nc > 64) nc = 64;
  if (nc < 1) nc = 1;
  if (cell_size) *cell_size = kCube / nc;
  return nc;
}

int shard_owner_of_cell(int wx, int wy, int wz, int cx, int cy, int cz, int world_size) {
  if (world_size <= 1) return 0;
  return (int)(brick_hash(wx, wy, wz, cx / kBrickCells, cy / kBrickCells, cz / kBrickCells) % (uint32_t)world_size);
}

LocalMap::LocalMap() {
  origin_[0] = (int)(kMapW * 0.5); origin_[1] = (int)(kMapH * 0.5); origin_[2] = (int)(kMapD * 0.5);  // LocalMap.h:141-144
}

void LocalMap::clear() {
  for (auto& c : cubes_) c.reset();
  version_++;
}

void LocalMap::set_origin(const double t[3]) {
  for (int a = 0; a < 3; ++a) origin_[a] = -cube_coord(t[a], 0);
  version_++;
}

bool LocalMap::shift(const double t[3], int pos[3]) {
  int ci = cube_coord(t[0], origin_[0]), cj = cube_coord(t[1], origin_[1]), ck = cube_coord(t[2], origin_[2]);
  bool moved = false;
  auto& M = cubes_;
  // Each while-loop rolls the block array by one along an axis so that the sensor's block stays
  // >= 3 blocks away from the border (LocalMap.h:183-284); the block leaving the window is dropped.
  while (ci < 3) {
    for (int j = 0; j < kMapH; ++j) for (int k = 0; k < kMapD; ++k) {
      for (int i = kMapW - 1; i >= 1; --i) M[cidx(i, j, k)] = std::move(M[cidx(i - 1, j, k)]);
      M[cidx(0, j, k)].reset();
    }
    ci++; origin_[0]++; moved = true;
  }
  while (ci >= kMapW - 3) {
    for (int j = 0; j < kMapH; ++j) for (int k = 0; k < kMapD; ++k) {
      for (int i = 0; i < kMapW - 1; ++i) M[cidx(i, j, k)] = std::move(M[cidx(i + 1, j, k)]);
      M[cidx(kMapW - 1, j, k)].reset();
    }
    ci--; origin_[0]--; moved = true;
  }
  while (cj < 3) {
    for (int i = 0; i < kMapW; ++i) for (int k = 0; k < kMapD; ++k) {
      for (int j = kMapH - 1; j >= 1; --j) M[cidx(i, j, k)] = std::move(M[cidx(i, j - 1, k)]);
      M[cidx(i, 0, k)].reset();
    }
    cj++; origin_[1]++; moved = true;
  }
  while (cj >= kMapH - 3) {
    for (int i = 0; i < kMapW; ++i) for (int k = 0; k < kMapD; ++k) {
      for (int j = 0; j < kMapH - 1; ++j) M[cidx(i, j, k)] = std::move(M[cidx(i, j + 1, k)]);
      M[cidx(i, kMapH - 1, k)].reset();
    }
    cj--; origin_[1]--; moved = true;
  }
  while (ck < 3) {
    for (int i = 0; i < kMapW; ++i) for (int j = 0; j < kMapH; ++j) {
      for (int k = kMapD - 1; k >= 1; --k) M[cidx(i, j, k)] = std::move(M[cidx(i, j, k - 1)]);
      M[cidx(i, j, 0)].reset();
    }
    ck++; origin_[2]++; moved = true;
  }
  while (ck >= kMapD - 3) {
    for (int i = 0; i < kMapW; ++i) for (int j = 0; j < kMapH; ++j) {
      for (int k = 0; k < kMapD - 1; ++k) M[cidx(i, j, k)] = std::move(M[cidx(i, j, k + 1)]);
      M[cidx(i, j, kMapD - 1)].reset();
    }
    ck--; origin_[2]--; moved = true;
  }
  pos[0] = ci; pos[1] = cj; pos[2] = ck;
  if (moved) version_++;
  return moved;
}

int LocalMap::cube_index_of(const float p[3]) const {
  const int ci = cube_coord((double)p[0], origin_[0]);
  const int cj = cube_coord((double)p[1], origin_[1]);
  const int ck = cube_coord((double)p[2], origin_[2]);
  if (!(ci >= 0 && ci < kMapW && cj >= 0 && cj < kMapH && ck >= 0 && ck < kMapD)) return -1;
  return cidx(ci, cj, ck);
}

// pcl::VoxelGrid<PointXYZI>::applyFilter restated: leaf coordinate = floor(v * inv_leaf) - min_b (float
// arithmetic), linear index x-fastest, centroids accumulated in float, emitted in ascending leaf index.
// Upstream's order INSIDE a leaf (boost spreadsort) is unspecified; we use the input order.
void LocalMap::voxel_grid(std::vector<float>& xyz, float leaf) {
  const size_t n = xyz.size() / 3;
  if (n == 0) return;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], xyz[3 * i + a]); mx[a] = std::max(mx[a], xyz[3 * i + a]); }
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) return;  // "Leaf size is too small ...": cloud passes through unfiltered
  int minb[3], divb[3];
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)std::floor(mn[a] * inv);
    divb[a] = (int)std::floor(mx[a] * inv) - minb[a] + 1;
  }
  std::vector<uint32_t> leaf_of(n), order(n);
  for (size_t i = 0; i < n; ++i) {
    const int i0 = (int)(std::floor(xyz[3 * i] * inv) - (float)minb[0]);
    const int i1 = (int)(std::floor(xyz[3 * i + 1] * inv) - (float)minb[1]);
    const int i2 = (int)(std::floor(xyz[3 * i + 2] * inv) - (float)minb[2]);
    leaf_of[i] = (uint32_t)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]);
  }
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return leaf_of[a] < leaf_of[b]; });
  std::vector<float> out;
  out.reserve(xyz.size());
  size_t i = 0;
  while (i < n) {
    size_t j = i;
    float s0 = 0, s1 = 0, s2 = 0;
    const uint32_t l = leaf_of[order[i]];
    while (j < n && leaf_of[order[j]] == l) {
      const float* p = &xyz[3 * (size_t)order[j]];
      s0 += p[0]; s1 += p[1]; s2 += p[2];
      ++j;
    }
    const float cnt = (float)(j - i);
    out.push_back(s0 / cnt); out.push_back(s1 / cnt); out.push_back(s2 / cnt);
    i = j;
  }
  xyz.swap(out);
}

int LocalMap::add_surf(const float* xyz, size_t n, size_t stride) {
  if (stride == 0) stride = 3;
  std::vector<uint8_t> touched(kMapNum, 0);
  int inserted = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* p = xyz + i * stride;
    const int ci = cube_index_of(p);  // LocalMap.h:596-610
    if (ci < 0) continue;
    if (!cubes_[ci]) cubes_[ci] = std::make_unique<Cube>();
    cubes_[ci]->xyz.insert(cubes_[ci]->xyz.end(), p, p + 3);
    touched[ci] = 1;
    inserted++;
  }
  for (int ci = 0; ci < kMapNum; ++ci)  // LocalMap.h:617-641 (tbb::parallel_for over touched blocks)
    if (touched[ci]) voxel_grid(cubes_[ci]->xyz, plane_res_);
  if (inserted) version_++;
  return inserted;
}

int LocalMap::count_5x5(const int pos[3]) const {
  int n = 0;
  for (int i = pos[0] - 2; i <= pos[0] + 2; ++i)
    for (int j = pos[1] - 2; j <= pos[1] + 2; ++j)
      for (int k = pos[2] - 1; k <= pos[2] + 1; ++k)
        if (i >= 0 && i < kMapW && j >= 0 && j < kMapH && k >= 0 && k < kMapD && cubes_[cidx(i, j, k)])
          n += (int)(cubes_[cidx(i, j, k)]->xyz.size() / 3);
  return n;
}

size_t LocalMap::size() const {
  size_t n = 0;
  for (const auto& c : cubes_) if (c) n += c->xyz.size() / 3;
  return n;
}

// cell coordinate of a point inside cube (ci,cj,ck): floor((p - cube_min)/cell), clamped
static inline void cell_of(const float* p, const double cmin[3], double inv_cell, int nc, int c[3]) {
  for (int a = 0; a < 3; ++a) {
    int v = (int)std::floor(((double)p[a] - cmin[a]) * inv_cell);
    c[a] = v < 0 ? 0 : (v >= nc ? nc - 1 : v);
  }
}

void LocalMap::build_canonical(int rank, int world, CanonicalMap& out) const {
  out.nc = cells_per_cube(plane_res_, &out.cell);
  const int nc = out.nc;
  const double inv_cell = 1.0 / out.cell;
  const size_t ncell = (size_t)nc * nc * nc;
  out.cube_slot.assign(kMapNum, -1);
  out.slot_cube.clear();
  out.xyzw.clear(); out.cell_start.clear();
  out.total_points = size();
  out.n_slots = 0;
  std::vector<uint32_t> cellid, counts, fill;
  std::vector<uint8_t> keep;
  for (int cube = 0; cube < kMapNum; ++cube) {
    const Cube* c = cubes_[cube].get();
    if (!c || c->xyz.empty()) continue;
    const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
    const int w[3] = {ci - origin_[0], cj - origin_[1], ck - origin_[2]};  // world cube id
    const double cmin[3] = {w[0] * kCube - kHalfCube, w[1] * kCube - kHalfCube, w[2] * kCube - kHalfCube};
    const size_t n = c->xyz.size() / 3;
    cellid.resize(n); keep.assign(n, 1);
    counts.assign(ncell + 1, 0);
    for (size_t i = 0; i < n; ++i) {
      int g[3];
      cell_of(&c->xyz[3 * i], cmin, inv_cell, nc, g);
      cellid[i] = (uint32_t)((g[2] * nc + g[1]) * nc + g[0]);
      if (world > 1) {
        // needed by `rank` iff some cell of the 3x3x3 neighbourhood lies in a brick owned by it
        bool need = false;
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = std::max(g[a] - 1, 0) / kBrickCells; hi[a] = std::min(g[a] + 1, nc - 1) / kBrickCells; }
        for (int bz = lo[2]; bz <= hi[2] && !need; ++bz)
          for (int by = lo[1]; by <= hi[1] && !need; ++by)
            for (int bx = lo[0]; bx <= hi[0] && !need; ++bx)
              need = shard_owner_of_cell(w[0], w[1], w[2], bx * kBrickCells, by * kBrickCells, bz * kBrickCells, world) == rank;
        keep[i] = need;
      }
      if (keep[i]) counts[cellid[i] + 1]++;
    }
    // NOTE: a cube keeps its slot (tree exists, LocalMap.h:506) even if this rank holds none of its points
    const uint32_t base = (uint32_t)out.n_points();
    for (size_t k = 0; k < ncell; ++k) counts[k + 1] += counts[k];
    const size_t kept = counts[ncell];
    out.xyzw.resize(4 * (size_t)(base + kept), 0.0f);
    fill.assign(counts.begin(), counts.end() - 1);
    for (size_t i = 0; i < n; ++i) {
      if (!keep[i]) continue;
      const uint32_t dst = base + fill[cellid[i]]++;
      out.xyzw[4 * (size_t)dst] = c->xyz[3 * i]; out.xyzw[4 * (size_t)dst + 1] = c->xyz[3 * i + 1]; out.xyzw[4 * (size_t)dst + 2] = c->xyz[3 * i + 2];
    }
    const size_t off = out.cell_start.size();
    out.cell_start.resize(off + ncell + 1);
    for (size_t k = 0; k <= ncell; ++k) out.cell_start[off + k] = base + counts[k];
    out.cube_slot[cube] = out.n_slots++;
    out.slot_cube.push_back(cube);
  }
}

size_t LocalMap::export_points(float* xyz, size_t cap, bool only_5x5, const int pos[3]) const {
  // canonical order of the single-rank layout (what the device holds when world_size == 1)
  CanonicalMap cm;
  build_canonical(0, 1, cm);
  size_t n = 0;
  for (int s = 0; s < cm.n_slots; ++s) {
    const int cube = cm.slot_cube[s];
    if (only_5x5) {
      const int ci = cube % kMapW, cj = (cube / kMapW) % kMapH, ck = cube / (kMapW * kMapH);
      if (std::abs(ci - pos[0]) > 2 || std::abs(cj - pos[1]) > 2 || std::abs(ck - pos[2]) > 1) continue;
    }
    const size_t ncell = (size_t)cm.nc * cm.nc * cm.nc;
    const uint32_t b = cm.cell_start[s * (ncell + 1)], e = cm.cell_start[s * (ncell + 1) + ncell];
    for (uint32_t i = b; i < e; ++i) {
      if (n < cap && xyz) { xyz[3 * n] = cm.xyzw[4 * (size_t)i]; xyz[3 * n + 1] = cm.xyzw[4 * (size_t)i + 1]; xyz[3 * n + 2] = cm.xyzw[4 * (size_t)i + 2]; }
      ++n;
    }
  }
  return n;
}

}  // namespace soicp
