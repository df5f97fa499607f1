"""GPU: the RCCL (`nccl`) code paths of nn/parallel.py and the sharded StepEngine, on the one GPU a test box has.  Every
multi-rank test of this suite runs over gloo (RCCL refuses two ranks on one device); a world of ONE rank is legal RCCL and
executes the same calls: process-group init with a device id, the all-reduce between two graph replays, max-reduce of the
bench timing, barrier, teardown.  No scaling claim follows from this -- it only removes "has never executed" from those
lines.  One child process, one launch of each."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_runs_the_sharded_step():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("STROTSS_DIST_BACKEND",)}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               STROTSS_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_world1_worker.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert r["backend"] == "nccl" and r["world"] == 1 and r["rank"] == 0
    assert r["shares"] is False and not r["identity"].endswith(":cpu")
    assert r["allreduce_identity"] is True
    assert r["aggregate"] == [3.5, 2.0]
    # deterministic tap adjoint + a one-rank sum = identity: the sharded step, eager and as graph | all-reduce | graph,
    # reproduces the plain single-process step bit for bit (losses, first gradients, variables after three steps)
    assert r["bitwise_equal_to_plain_eager"] == {"plain/eager": True, "plain/graph": True, "sharded/eager": True,
                                                 "sharded/graph": True}, r
