// Known-answer generator for B8 (feature_selector.cpp:380-459): runs the REFERENCE's own 1-NN - the vendored, header-only
// nanoflann 1.3.0 (vins_estimator/lib/nanoflann/nanoflann.hpp), with the dataset adaptor, tree type, leaf size and search
// parameters of feature_selector.h:118-146 and feature_selector.cpp:424-455 - on seeded point clouds, and prints, per query,
// the index nanoflann returned and its squared distance.  It is the one piece of the reference that compiles in this image
// (STL only); nothing of it is copied into the repository: gen_nanoflann_nn.py compiles THIS file against the header where it
// lies under /root/reference and stores the numbers as tests/golden/nanoflann_nn.npz.
//
// Clouds: sizes 0 (findNNDepth's "return 1.0" branch), 1, 9, 10, 11 (around the leaf size), 64, 150 (a full window), twice each.
// Queries per cloud: random points of the normalized image plane, every cloud point itself (distance 0), and EXACT TIES -
// part of every cloud sits on a grid of dyadic coordinates (multiples of 1/64), and the tie queries are the midpoints of
// horizontally adjacent grid points and the centres of grid cells: two resp. four cloud points at bit-identical squared
// distances (all operands are dyadic rationals, the arithmetic is exact).
#include <cstdint>
#include <cstdio>
#include <limits>
#include <memory>
#include <random>
#include <utility>
#include <vector>

#include "nanoflann.hpp"

// feature_selector.h:118-141
struct PointCloud {
  PointCloud(const std::vector<std::pair<double, double>>& dataset) : pts(dataset) {}
  const std::vector<std::pair<double, double>>& pts;
  inline size_t kdtree_get_point_count() const { return pts.size(); }
  inline double kdtree_get_pt(const size_t idx, const size_t dim) const { return dim == 0 ? pts[idx].first : pts[idx].second; }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const { return false; }
};
// feature_selector.h:143
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, PointCloud>, PointCloud, 2> my_kd_tree_t;

// one cloud: print it, build the reference's index over it (feature_selector.cpp:424-429) and answer the queries (:447-457)
static void run_cloud(const std::vector<std::pair<double, double>>& dataset, const std::vector<double>& depths,
                      const std::vector<std::pair<double, double>>& queries);

// Second set (`gen_nanoflann_nn 2` -> nanoflann_nn2.npz): the degenerate clouds, where the tree's shape is decided by planeSplit's
// handling of equal coordinates and the answer by the traversal order alone - exact duplicates of points, clouds of ONE repeated
// point (middleSplit_'s "split in the middle to keep the tree balanced"), collinear clouds (one coordinate constant), a complete
// dyadic grid (every cell centre a four-way tie, every edge midpoint a two-way tie), a cloud deeper than a window's (300 points),
// queries far outside the root bounding box (computeInitialDistances), and queries with NaN / infinite coordinates (nothing is
// ever "closer": ret_index keeps its initial 0).
static int second_set() {
  std::mt19937_64 rng(0x5EC0DDull);
  std::uniform_real_distribution<double> ux(-0.8, 0.8), uy(-0.5, 0.5), ud(2.0, 15.0);
  const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
  typedef std::vector<std::pair<double, double>> pts_t;
  std::vector<pts_t> clouds;
  {  // 0: 40 random points, each three times (120)
    pts_t c;
    for (int i = 0; i < 40; i++) c.push_back({ux(rng), uy(rng)});
    pts_t d;
    for (int r = 0; r < 3; r++) d.insert(d.end(), c.begin(), c.end());
    clouds.push_back(d);
  }
  for (int n : {11, 23, 64, 150}) clouds.push_back(pts_t(n, {0.125, -0.25}));  // 1-4: one point, n times
  {  // 5, 6: collinear (x constant / y constant), dyadic steps with repeats
    pts_t a, b;
    for (int i = 0; i < 90; i++) a.push_back({0.25, ((int)(rng() % 33) - 16) / 32.0}), b.push_back({((int)(rng() % 41) - 20) / 32.0, -0.125});
    clouds.push_back(a), clouds.push_back(b);
  }
  {  // 7: a complete 12 x 10 dyadic grid (120), shuffled
    pts_t g;
    for (int y = 0; y < 10; y++)
      for (int x = 0; x < 12; x++) g.push_back({(x - 6) / 16.0, (y - 5) / 16.0});
    clouds.push_back(g);
  }
  {  // 8: 300 points, a third of them on a coarse dyadic lattice (many duplicates)
    pts_t c;
    for (int i = 0; i < 200; i++) c.push_back({ux(rng), uy(rng)});
    for (int i = 0; i < 100; i++) c.push_back({((int)(rng() % 9) - 4) / 8.0, ((int)(rng() % 7) - 3) / 8.0});
    clouds.push_back(c);
  }
  {  // 9: 150 points with only four distinct x values (the split value lands on a coordinate: the == cutval band of planeSplit)
    pts_t c;
    for (int i = 0; i < 150; i++) c.push_back({((int)(rng() % 4)) / 4.0 - 0.5, uy(rng)});
    clouds.push_back(c);
  }
  std::printf("%d\n", (int)clouds.size());
  for (size_t ci = 0; ci < clouds.size(); ci++) {
    pts_t& dataset = clouds[ci];
    const int n = (int)dataset.size();
    for (int i = n - 1; i > 0; i--) std::swap(dataset[i], dataset[rng() % (i + 1)]);
    std::vector<double> depths;
    for (int i = 0; i < n; i++) depths.push_back(ud(rng));
    pts_t queries;
    for (int q = 0; q < 300; q++) queries.push_back({ux(rng), uy(rng)});
    for (int i = 0; i < n; i++) queries.push_back(dataset[i]);
    for (int q = 0; q < 200; q++) queries.push_back({((int)(rng() % 65) - 32) / 32.0, ((int)(rng() % 65) - 32) / 32.0});  // dyadic: ties with the lattices
    for (int q = 0; q < 24; q++) queries.push_back({(q % 2 ? 50.0 : -50.0) * (1 + q % 3), (q % 4 < 2 ? 30.0 : -0.25) * (1 + q % 5)});  // far outside
    queries.push_back({nan, 0.0}), queries.push_back({0.0, nan}), queries.push_back({nan, nan});
    queries.push_back({inf, 0.0}), queries.push_back({-inf, 0.25}), queries.push_back({0.0, inf}), queries.push_back({inf, -inf});
    run_cloud(dataset, depths, queries);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == '2') return second_set();
  std::mt19937_64 rng(0xB8B8B8ull);
  std::uniform_real_distribution<double> ux(-0.8, 0.8), uy(-0.5, 0.5), ud(2.0, 15.0);
  const int sizes[] = {0, 1, 9, 10, 11, 64, 150, 0, 1, 9, 10, 11, 64, 150};
  std::printf("%d\n", (int)(sizeof(sizes) / sizeof(sizes[0])));
  for (int n : sizes) {
    std::vector<std::pair<double, double>> dataset;
    std::vector<double> depths;
    // a 4 x G grid of dyadic points (as many as fit into a third of the cloud), then random points
    const int G = n >= 12 ? (n / 3) / 4 : 0;
    for (int gy = 0; gy < (G ? 4 : 0); gy++)
      for (int gx = 0; gx < G; gx++) dataset.push_back({(gx - G / 2) * (8.0 / 64.0), (gy - 2) * (8.0 / 64.0)});
    while ((int)dataset.size() < n) dataset.push_back({ux(rng), uy(rng)});
    // (list order = f_manager.feature order; shuffle so that the grid points are not the low indices)
    for (int i = n - 1; i > 0; i--) std::swap(dataset[i], dataset[rng() % (i + 1)]);
    for (int i = 0; i < n; i++) depths.push_back(ud(rng));
    std::vector<std::pair<double, double>> queries;
    for (int q = 0; q < 700; q++) queries.push_back({ux(rng), uy(rng)});
    for (int i = 0; i < n; i++) queries.push_back(dataset[i]);
    for (int gy = 0; gy < (G ? 4 : 0); gy++)
      for (int gx = 0; gx + 1 < G; gx++) {
        const double x0 = (gx - G / 2) * (8.0 / 64.0), y0 = (gy - 2) * (8.0 / 64.0);
        queries.push_back({x0 + 4.0 / 64.0, y0});                            // between two grid points
        if (gy + 1 < 4) queries.push_back({x0 + 4.0 / 64.0, y0 + 4.0 / 64.0});  // centre of a cell: four at the same distance
      }
    run_cloud(dataset, depths, queries);
  }
  return 0;
}

static void run_cloud(const std::vector<std::pair<double, double>>& dataset, const std::vector<double>& depths,
                      const std::vector<std::pair<double, double>>& queries) {
  const int n = (int)dataset.size();
  std::printf("%d %d\n", n, (int)queries.size());
  for (int i = 0; i < n; i++) std::printf("%.17g %.17g %.17g\n", dataset[i].first, dataset[i].second, depths[i]);
  // feature_selector.cpp:424-429
  PointCloud cloud(dataset);
  std::unique_ptr<my_kd_tree_t> kdtree;
  if (n > 0) {
    kdtree.reset(new my_kd_tree_t(2, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10)));
    kdtree->buildIndex();
  }
  for (const auto& q : queries) {
    long long idx = -1;
    double out_dist_sqr = 0.0, depth = 1.0;  // feature_selector.cpp:444: an empty cloud answers 1.0
    if (n > 0) {
      // feature_selector.cpp:447-457
      double query_pt[2] = {q.first, q.second};
      const size_t num_results = 1;
      size_t ret_index = 0;
      nanoflann::KNNResultSet<double> resultSet(num_results);
      resultSet.init(&ret_index, &out_dist_sqr);
      kdtree->findNeighbors(resultSet, &query_pt[0], nanoflann::SearchParams(10));
      idx = (long long)ret_index;
      depth = depths[ret_index];
    }
    std::printf("%.17g %.17g %lld %.17g %.17g\n", q.first, q.second, idx, out_dist_sqr, depth);
  }
}
