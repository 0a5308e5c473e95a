import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from superodom_amd import binding, synth
sc = synth.Scene("os1_128_2m")
slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4, max_surface_features=-1)
slam.add_surf_point_cloud(sc.map_points)
for r in range(6):
    i = r % 4
    t = time.perf_counter()
    rc, pose, st = slam.register(sc.scan(i), sc.guess(i))
    dt = time.perf_counter() - t
    print("HAND", os.environ.get("SOICP_KNN_HAND"), "scan", i, "rc", rc, "ms %.3f" % (1e3 * dt), "outer", st.n_iterations, "handed over", st.knn_handed_over, "err", synth.pose_error(pose, sc.gt_pose(i)))
