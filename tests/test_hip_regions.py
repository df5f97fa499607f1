"""GPU: mask regions sharded over ranks (BASELINE config 4; reference semantics run_strotss.py:104-125: the loss is
the mean of R region losses sharing one VGG pass).  Two real processes with a real collective (gloo on this box's single
GPU; RCCL refuses two ranks on one device) must reproduce the single-process masked step: every rank runs the replicated
trunk, its own regions' losses and their data-gradient, ONE all-reduce sums [pixel gradient | scalars]."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_region_worker.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _two_ranks(tmp_path, mode):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), STROTSS_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, WORKER, str(tmp_path / "out"), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    return [torch.load(f"{tmp_path / 'out'}.r{r}.pt") for r in range(2)]


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_region_sharded_step_equals_single_process(tmp_path, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _region_worker as W
    eng, idx = W.problem(torch.device("cuda", 0), None)
    assert eng.world == 1 and eng.my_regions == [0, 1, 2]
    ref = W.run(eng, idx, use_graph=(mode == "graph"))
    r0, r1 = _two_ranks(tmp_path, mode)
    assert r0["my_regions"] == [0, 2] and r1["my_regions"] == [1]
    assert r0["two_graphs"] == r1["two_graphs"] == (mode == "graph")
    for k in ("loss", "loss_c", "loss_s"):
        # every rank logs the all-reduced scalars of ALL regions; they equal the single-process ones
        assert abs(r0["losses0"][k] - ref["losses0"][k]) < 2e-5 * max(1.0, abs(ref["losses0"][k])), (k, r0["losses0"], ref["losses0"])
        assert r0["losses0"][k] == r1["losses0"][k]
    for a, b, c in zip(ref["gvars0"], r0["gvars0"], r1["gvars0"]):
        assert torch.equal(b, c)                     # identical on both ranks -> identical replicated update
        rel = float((a - b).norm() / a.norm())
        assert rel < 1e-4, rel                       # same kernels, same inputs: only the summation order differs
    for b, c in zip(r0["variables"], r1["variables"]):
        assert torch.equal(b, c)                     # three steps later the ranks still hold the same image
    # the trajectories: sign-like first RMSprop steps amplify rounding noise, so compare the losses
    assert abs(r0["losses2"]["loss"] - ref["losses2"]["loss"]) < 2e-2 * abs(ref["losses2"]["loss"])


def test_bench_regions_mode_two_ranks_prints_n_gpus_2():
    """`python bench.py --gpus 2 --mode regions` with no torchrun around it: the script starts its two ranks itself
    (gloo rehearsal on one GPU) and rank 0 prints ONE JSON line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["STROTSS_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "regions", "--scale", "128",
                          "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "4 mask regions" in line["config"]["workload"] and "all-reduce" in line["config"]["parallelism"]
