"""`python bench.py --gpus N` must start the way the driver starts it -- with or without a launcher around it -- and leave exactly
ONE JSON line on stdout (VERDICT r04 "Next round" 3a).  CPU only: --dry-control-plane runs everything around the registrations
(self-launch through torch.distributed.run on 127.0.0.1, gloo rendezvous, barrier, max-over-ranks, rank-0 print)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    return r.returncode, lines, r.stderr.decode()


@pytest.mark.parametrize("n", [1, 2])
def test_bench_self_launch_prints_one_json_line(n):
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--dry-control-plane"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["dry_control_plane"] is True and d["n_gpus"] == n and d["max_over_ranks_of_rank_plus_1"] == float(n)


def test_bench_under_external_launcher():
    port = 29600 + os.getpid() % 300
    rc, lines, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-control-plane"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    assert json.loads(lines[0])["n_gpus"] == 2
