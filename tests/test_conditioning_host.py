"""The product's LM controller (lm_solver.h: Cholesky on the Jacobi-scaled normal equations) against the oracle's (QR on the stacked
Jacobian, the route Ceres DENSE_QR takes) on RANK-DEFICIENT geometry, where normal equations square the condition number:
equal counters and termination codes, poses to 1e-9.  CPU only (the device runs the same header; tests/test_gpu_conditioning.py
puts the whole HIP path on the same scenes)."""
import numpy as np
import pytest

from helpers import DegenerateScene
from superodom_amd import synth


@pytest.mark.parametrize("name,sigma", [("floor_only", 0.01), ("open_corridor", 0.01), ("two_walls", 0.01), ("floor_only", 0.003)])
def test_host_lm_equals_qr_oracle_on_rank_deficient_scenes(oracle, soicp, name, sigma):
    sc = DegenerateScene(name, sigma=sigma)
    om = oracle.OracleMap(plane_res=sc.plane_res)
    om.add_surf(sc.map_points)
    for i in range(2):
        scan, guess = sc.scan(i), sc.guess(i)
        corrs = om.plane_match(guess, scan)
        pose_o, st_o = oracle.lm_solve(corrs, guess, sc.plane_res)

        def sums_at(x):
            cost, JtJ, Jtr, cnt = oracle.evaluate(corrs, x, sc.plane_res)
            return soicp.LmDriver.sums(cost, cnt, Jtr, JtJ)
        s0 = sums_at(guess)
        H = np.zeros((6, 6)); k = 0
        for a in range(6):
            for b in range(a, 6):
                H[a, b] = H[b, a] = s0.JtJ[k]; k += 1
        ev = np.linalg.eigvalsh(H)
        cond = ev[-1] / max(ev[0], 1e-300)
        print(f"{name} sigma {sigma} scan {i}: cond(JtJ) = {cond:.3e}, accepted {int(s0.count)}")
        assert cond > 5e3, "the scene is meant to be ill conditioned"
        drv = soicp.LmDriver()
        more, nxt = drv.begin(guess, s0, 4)
        while more:
            more, nxt = drv.feed(sums_at(nxt))
        pose_p, st_p = drv.result()
        assert (st_p.lm_iterations, st_p.num_successful_steps, st_p.termination) == (st_o.lm_iterations, st_o.num_successful_steps, st_o.termination)
        dt, dr = synth.pose_error(pose_p, pose_o)
        assert dt < 1e-9 and dr < 1e-9, (dt, dr)
