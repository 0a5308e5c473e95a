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

NAMES = ["loop", "lds_reduce+store", "ticket", "load_partials", "reduce+sums", "lm", "total", "controller"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="os1_128_2m")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--scan", type=int, default=0, help="scan (and guess) of the trajectory registered by the plain loop")
    ap.add_argument("--staged", action="store_true", help="the bench's stream: four scans in rotation, each announced during the registration before it "
                                                           "(so_icp_stage_scan from pinned memory) -- the sweeps then run on chunks binned AHEAD, under the previous guess")
    a = ap.parse_args()
    sc = synth.Scene(a.workload)
    slam = binding.LidarSlamGpu(device_id=0, plane_res=sc.plane_res, line_res=sc.plane_res / 2, max_iterations=5,
                                lm_max_iterations=4, max_surface_features=-1)
    slam.add_surf_point_cloud(sc.map_points)
    d = slam.upload_scan(sc.scan(a.scan))
    st = binding.Stats()
    acc = np.zeros((2, 8))
    lm_stamps = os.environ.get("SOICP_LM_STAMPS")  # library built with -DSO_LM_STAMPS: dbg[0..7] = clocks inside the controller (slot 1)
    lm_acc = np.zeros(7)
    host = [slam.host_alloc_like(np.ascontiguousarray(sc.scan(i), dtype=np.float32)) for i in range(4)] if a.staged else None
    if a.staged:
        slam.stage_scan(host[0])
    for r in range(a.reps + 2):
        if a.staged:  # (the last registration of the loop is scan (reps + 1) % 4, binned under the guess of the one before)
            slam.stage_scan(host[(r + 1) % 4])
            _, _, st = slam.register(host[r % 4], sc.guess(r % 4))
            ahead = bool(st.flags & binding.FLAG_BINNED_AHEAD)
        else:
            slam.register_dev(d[0], d[1], sc.guess(a.scan), st)
        s = slam.debug_stamps().astype(np.float64) * 0.01  # 100 MHz -> us
        if r >= 2 and lm_stamps:
            lm_acc += np.diff(slam.debug_stamps().astype(np.float64)[0:8]) * 0.01
        if r >= 2:
            acc[0] += s[0:8]
            acc[1] += s[8:16]
    acc /= a.reps
    if a.staged:
        print('last registration binned ahead:', ahead)
    if lm_stamps:
        print("controller phases (us):", dict(zip(["feed:tolerances+rel", "feed:accept+unpack+gradient", "propose:scale+Hs", "cholesky+solve", "model change",
                                                    "pose_plus", "store back"], np.round(lm_acc / a.reps, 2))))
    rec = slam.debug_knn_stamps()
    if rec is not None:
        steals_all = (rec[..., 12].astype(np.uint64) >> np.uint64(32)).astype(np.int64)  # queries a wavefront claimed from the hand-over ring
        rec = rec.copy(); rec[..., 12] = rec[..., 12].astype(np.uint64) & np.uint64(0xFFFFFFFF)
        rec = rec.astype(np.float64)
        for o in range(2):
            r = rec[o]
            stl = steals_all[o][r[:, 0] > 0]
            r = r[r[:, 0] > 0]
            if len(r) and stl.sum():
                t0_ = r[:, 0].min()
                w_ = stl > 0
                print(f"knn sweep {o}: hand-over ring: {int(stl.sum())} queries claimed by {int(w_.sum())} wavefronts (max {int(stl.max())} by one); "
                      f"those wavefronts end at p50 {np.percentile((r[w_, 1] - t0_) * 0.01, 50):.1f} max {((r[w_, 1] - t0_) * 0.01).max():.1f} us, the others at max {((r[~w_, 1] - t0_) * 0.01).max():.1f} us")
            if not len(r):
                continue
            t0 = r[:, 0].min()
            busy = r[r[:, 7] > 0]
            nch = busy[:, 7].sum()
            ph = busy[:, 2:7].sum(axis=0) * 0.01 / nch
            print(f"knn sweep {o}: span {(r[:, 1].max() - t0) * 0.01:.1f} us; wavefronts {len(r)} ({len(busy)} with work), chunks {int(nch)}, "
                  f"queries {int(busy[:, 10].sum())}, candidates/chunk {busy[:, 9].sum() / nch:.0f}, groups/chunk {busy[:, 11].sum() / nch:.2f}, "
                  f"chunks with a full pass {int(busy[:, 12].sum())}")
            print("   per chunk (us): prologue %.2f  group_setup %.2f  stage+scan %.2f  merge+survivors %.2f  epilogue %.2f" % tuple(ph))
            span = busy[:, 1] - busy[:, 0]
            print("   shader clock over the wavefronts' lives: median %.0f MHz" % (np.median(busy[:, 15] / np.maximum(span, 1)) * 100.0))
            start = (busy[:, 0] - t0) * 0.01
            life = (busy[:, 1] - busy[:, 0]) * 0.01
            print("   workgroup start offset us: p50 %.1f p90 %.1f max %.1f | life us: p50 %.1f p90 %.1f max %.1f | max chunk %.1f us" % (
                np.percentile(start, 50), np.percentile(start, 90), start.max(), np.percentile(life, 50), np.percentile(life, 90), life.max(),
                busy[:, 8].max() * 0.01))
            k = int(np.argmax(busy[:, 8]))
            mi = int(busy[k, 13])
            print("   slowest chunk: %.1f us, candidates scanned %d, groups(all passes) %d, fallback lanes %d | fallback lanes total %d" % (
                busy[k, 8] * 0.01, mi >> 32, mi & 0xFFFF, (mi >> 16) & 0xFFFF, int((busy[:, 14].astype(np.uint64) & np.uint64(0xFFFFFFFF)).sum())))
            order = np.argsort(-busy[:, 8])[:8]
            print("   top chunks (us, cand, fb):", [(round(busy[q, 8] * 0.01, 1), int(busy[q, 13]) >> 32, (int(busy[q, 13]) >> 16) & 0xFFFF) for q in order])
            if os.environ.get("SOICP_HWID"):  # where the wavefronts ran: (xcc, se, cu, simd) from HW_ID / XCC_ID in d[14]
                hw = (r[:, 14].astype(np.uint64) >> np.uint64(32))
                xcc = (hw >> np.uint64(16)) & np.uint64(0xF); simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF)
                se = (hw >> np.uint64(13)) & np.uint64(7)
                for w in list(range(0, 12)) + [1024, 1025, 2048, 2049, 3072, 3073, 4092, 4093]:
                    if w < len(r):
                        print("      wave %4d (wg %4d): xcc %d se %d cu %2d simd %d" % (w, w // 4, xcc[w], se[w], cu[w], simd[w]))
                place = (xcc * np.uint64(4096) + se * np.uint64(256) + cu * np.uint64(4) + simd).astype(np.int64)
                cost = r[:, 8] * 0.01
                u, inv = np.unique(place, return_inverse=True)
                loads = np.bincount(inv, weights=cost); cnts = np.bincount(inv)
                print("      distinct SIMDs %d; wavefronts per SIMD min %d max %d; chunk-time per SIMD: mean %.1f max %.1f us" % (len(u), cnts.min(), cnts.max(), loads.mean(), loads.max()))
            single = busy[busy[:, 7] == 1]
            if len(single):  # phase breakdown of the slowest single-chunk wavefronts: (total | prologue, set-up, stage+scan, merge, epilogue | queries, candidates, start)
                so = np.argsort(-single[:, 8])[:6]
                print("   slowest single-chunk wavefronts:")
                for q in so:
                    w = single[q]
                    print("      %.1f us | %s | queries %d cand %d start %.1f" % (w[8] * 0.01, " ".join("%.1f" % (x * 0.01) for x in w[2:7]), int(w[10]), int(w[9]), (w[0] - t0) * 0.01))
            idle = r[r[:, 7] == 0]
            if len(idle):
                print("   idle workgroups: start offset p50 %.1f max %.1f us" % (np.percentile((idle[:, 0] - t0) * 0.01, 50), ((idle[:, 0] - t0) * 0.01).max()))
    for k, row in zip(("fit ", "eval"), acc):
        print(k, {n: round(float(v), 2) for n, v in zip(NAMES, row)})


if __name__ == "__main__":
    main()
