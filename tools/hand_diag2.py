import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from superodom_amd import binding, synth
name = sys.argv[1] if len(sys.argv) > 1 else "small"
sc = synth.Scene(name)
def mk(h):
    os.environ["SOICP_KNN_HAND"] = str(h)
    s = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=1, lm_max_iterations=4, max_surface_features=-1)
    s.add_surf_point_cloud(sc.map_points)
    return s
a, b = mk(0), mk(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
for i in range(4):
    scan, guess = sc.scan(i), sc.guess(i)
    ra, pa, sa = a.register(scan, guess); ma = a.match_status(len(scan)).copy(); na = a.neighbours(len(scan))
    for rep in range(3):
        rb, pb, sb = b.register(scan, guess); mb = b.match_status(len(scan)).copy(); nb = b.neighbours(len(scan))
        d = np.flatnonzero(ma != mb)
        print(f"scan {i} rep {rep}: handed {sb.knn_handed_over}, status bytes differing {len(d)} {[(int(k), int(ma[k]), int(mb[k])) for k in d[:8]]}, pose equal {np.array_equal(pa, pb)}, "
              f"hist0 {list(sa.iterations[0].reject_hist)} histH {list(sb.iterations[0].reject_hist)}")
        pend = (ma == 255) & (mb == 255)
        ld = np.flatnonzero(pend & (na != nb).any(axis=1))
        print(f"    PENDING in both: {int(pend.sum())}, lists differing: {len(ld)}", [(int(k), na[k].tolist(), nb[k].tolist()) for k in ld[:4]])
        for k in d[:5]:
            print("    query", int(k), "lists", na[k].tolist(), nb[k].tolist())
