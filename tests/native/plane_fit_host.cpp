// Host build of superodom_amd/csrc/plane_fit.h (the product's plane fit is host + device code) for the CPU suite:
// tests/test_plane_fit_host.py drives it through ctypes.  Built on demand by the test (g++ -O2 -ffp-contract=off).
#include "plane_fit.h"

using namespace soicp;

extern "C" void pf_fit(const float* nb, const double* pw, const double* pose7, float sq_max_dist_f, double max_point_dist, int n,
                       int obs_as_written, double* nd /*4n*/, double* coeff /*n*/, int* status /*n*/, int* obs /*3n*/) {
  const Pose pose = pose_from_array(pose7);
  const ObsAxes ax = obs_axes(pose);
  for (int i = 0; i < n; ++i) {
    double o[4] = {0, 0, 0, 0}, c = 0;
    int ob[3] = {-1, -1, -1};
    status[i] = plane_fit5(nb + 15 * (size_t)i, pw + 3 * (size_t)i, ax, sq_max_dist_f, max_point_dist, o, c, ob, obs_as_written != 0);
    for (int k = 0; k < 4; ++k) nd[4 * (size_t)i + k] = o[k];
    coeff[i] = c;
    for (int k = 0; k < 3; ++k) obs[3 * (size_t)i + k] = ob[k];
  }
}
