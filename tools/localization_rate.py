#!/usr/bin/env python3
"""Localization() end to end at BASELINE sizes (registration + device-side map insert): wall time per call.
usage (GPU box): python tools/localization_rate.py [--calls 64] [--modes default,nodefer,hostbuilt]
  default    the insert is laid out by the device and completes behind the call (device_map.h: insert_fast / settle)
  nodefer    SOICP_MAP_FAST=sync: the call waits for the insert's report
  hostbuilt  SOICP_MAP_FAST=0: every insert round laid out by the host (two read-backs per insert; rounds 1-3)
  staged     default switches, and the NEXT scan is announced before every call (so_icp_stage_scan, scans in so_icp_host_alloc memory):
             its upload runs beside the registration instead of behind the insert -- what a node does whose feature callback
             hands the scan over as soon as it has it
The environment switches are read when a context is created, so one process measures all of them on the same scene."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superodom_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=64); ap.add_argument("--modes", default="default")
a = ap.parse_args()
# node / node_hostbuilt: what laserMapping does per frame (lmap.cpp:600-651, then :250-263): the surf cloud is voxel-filtered at
# planeRes on the device (so_icp_prefilter_scan) and Localization() runs on the filtered cloud, which is also what is inserted
# node_r3: the node order with this session's switches off (host-built insert rounds, host-decided pre-filter in the context's queue)
ENV = {"node": {}, "node_pageable": {}, "node_hostbuilt": {"SOICP_MAP_FAST": "0"},
       "node_r3": {"SOICP_MAP_FAST": "0", "SOICP_PREFILTER_FAST": "0"}, "default": {}, "nodefer": {"SOICP_MAP_FAST": "sync"}, "hostbuilt": {"SOICP_MAP_FAST": "0"}, "staged": {}, "staged_hostbuilt": {"SOICP_MAP_FAST": "0"}}
sc = synth.Scene("os1_128_2m")
scans = [sc.scan(i % 4) for i in range(4)]; guesses = [sc.guess(i % 4) for i in range(4)]
for mode in a.modes.split(","):
    for k in ("SOICP_MAP_FAST", "SOICP_PREFILTER_FAST"):
        os.environ.pop(k, None)
    os.environ.update(ENV[mode])
    slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5, lm_max_iterations=4,
                                max_surface_features=-1)
    slam.add_surf_point_cloud(sc.map_points)
    slam.shift_map(sc.gt_pose(0)[:3])
    staged = mode.startswith("staged")
    # (node: the raw clouds sit in so_icp_host_alloc memory, like the message pool of INTEGRATION.md section 6; node_pageable / node_r3: numpy's)
    bufs = [slam.host_alloc_like(x) for x in scans] if (staged or mode in ("node", "node_samequeue", "node_hostbuilt")) else scans
    times = []; pre = []; nf = 0
    t_all = time.perf_counter()
    for k in range(a.calls + 2):
        if k == 2:
            slam.map_size(); t_all = time.perf_counter()  # (settled: the clock starts on an idle device)
        i = k % 4
        t = time.perf_counter()
        if staged:
            if k == 0:
                slam.stage_scan(bufs[0])
            slam.stage_scan(bufs[(k + 1) % 4])
        if mode.startswith("node"):
            d_scan, n_f, info = slam.prefilter_scan(bufs[i], False, sc.plane_res / 2, sc.plane_res)
            if mode == "node":  # (the feature callback has the next raw cloud long before process() reaches it: its copy overlaps this frame's registration)
                slam.prefilter_announce(bufs[(k + 1) % 4])
            t_pre = time.perf_counter() - t
            rc, pose, st = slam.localization_dev(True, guesses[i], d_scan, n_f, 0.1 * k)
            pre.append(t_pre); nf = n_f
        else:
            rc, pose, st = slam.localization(True, guesses[i], bufs[i], 0.1 * k)
        times.append(time.perf_counter() - t)
        assert rc == 0
    n_map = slam.map_size()  # (waits for the last insert: the period below includes every insert)
    period = (time.perf_counter() - t_all) / a.calls * 1e3
    t = np.array(times[2:]) * 1e3
    print("[%s] Localization() ms per call: mean %.3f median %.3f min %.3f max %.3f | period incl. the inserts %.3f | registration part (time_elapsed_ms of the last call) %.3f | "
          "map size %d | inserts laid out by the device / handed back %s" % (mode, t.mean(), float(np.median(t)), t.min(), t.max(), period, st.time_elapsed_ms, n_map,
                                                                        slam.map_insert_stats()) +
          (" | of which prefilter (upload + VoxelGrid) %.3f, filtered scan %d points" % (np.mean(pre[2:]) * 1e3, nf) if pre else ""))
    slam.close()
