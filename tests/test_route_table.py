"""CPU: the DEFAULT kernel routing of every generic VGG16 layer (forward and data-gradient) at the five BASELINE scales is
pinned to tests/golden/route_table.json.  The routing is a size policy split between nn/model.py (`winograd_tile`,
`direct_splitk`) and the library (`strotss_conv3x3_winograd_route`, `strotss_conv3x3_workspace_bytes`) with ~25 STROTSS_*
switches and thresholds tuned by A/B runs on single boxes: without this pin a policy regression would pass every parity
test (all routes compute the same convolution, model.py:44-55 of the reference).  Runs in a child process with the
STROTSS_* variables removed (they are read once per process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(env_extra=None):
    env = {k: v for k, v in os.environ.items() if not k.startswith("STROTSS_")}
    env.update(env_extra or {})
    code = ("import json, sys; sys.path.insert(0, %r); import make_route_table as T; print(json.dumps(T.table()))"
            % os.path.join(ROOT, "tests", "golden"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_default_route_table_is_the_committed_one():
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "route_table.json")))
    got = _table()
    assert sorted(got) == sorted(want) == ["1024", "128", "256", "512", "64"]
    for scale in want:
        for w, g in zip(want[scale], got[scale]):
            assert w == g, (scale, w, g)
        assert len(want[scale]) == len(got[scale]) == 12
    # what DESIGN.md 4 says about the 1024-px scale: 8 launches of the fused kernel per step, block5 on 64 x 64 tiles
    fused = sum(r[5:].count("F4_fused_f32") for r in got["1024"])
    assert fused == 8 and all(r[5] == r[6] == "F4_x3_gemm_64" for r in got["1024"] if r[0].startswith("block5"))
    assert all(r[5] == r[6] == "direct_splitk" for r in got["64"])


def test_route_switches_move_the_table():
    """The pin is not vacuous: the documented switches change the routes."""
    off = _table({"STROTSS_X3": "0"})
    assert not any("x3" in r[5] or "x3" in r[6] for rows in off.values() for r in rows)
    nowino = _table({"STROTSS_WINOGRAD": "0"})
    assert all(r[5].startswith("direct") and r[6].startswith("direct") for rows in nowino.values() for r in rows)
    nofused = _table({"STROTSS_WINO_FUSED": "0"})
    assert not any("fused" in r[5] or "fused" in r[6] for rows in nofused.values() for r in rows)
