#!/usr/bin/env python3
"""Phase breakdown of eval_kernel (wall_clock64 stamps written by the last workgroup; SOICP_ABLATE bit 128).

Usage on a GPU box:  SOICP_ABLATE=128 python tools/eval_stamps.py [--workload os1_128_2m]
Prints microseconds per phase for the FIT (slot 0) and the plain evaluation launch, averaged over registrations.
wall_clock64 ticks at 100 MHz on gfx950 (10 ns).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SOICP_ABLATE", "128")

from superodom_amd import binding, synth  # noqa: E402

NAMES = ["loop", "lds_reduce+store", "ticket", "load_partials", "reduce+sums", "lm", "total"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="os1_128_2m")
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    sc = synth.Scene(a.workload)
    slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5,
                                lm_max_iterations=4, max_surface_features=-1)
    slam.add_surf_point_cloud(sc.map_points)
    d = slam.upload_scan(sc.scan(0))
    st = binding.Stats()
    acc = np.zeros((2, 7))
    for r in range(a.reps + 2):
        slam.register_dev(d[0], d[1], sc.guess(0), st)
        s = slam.debug_stamps().astype(np.float64) * 0.01  # 100 MHz -> us
        if r >= 2:
            acc[0] += s[0:7]
            acc[1] += s[8:15]
    acc /= a.reps
    for k, row in zip(("fit ", "eval"), acc):
        print(k, {n: round(float(v), 2) for n, v in zip(NAMES, row)})


if __name__ == "__main__":
    main()
