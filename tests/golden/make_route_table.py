"""Writes tests/golden/route_table.json: which kernels every generic VGG16 layer runs, forward and data-gradient, at the five
BASELINE scales under the DEFAULT policy (no STROTSS_* switches set).  Regenerate deliberately when a routing threshold is
changed on a measurement, and say so in the commit: tests/test_route_table.py compares against this file."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "strotss-tensorflow_amd")]
from nn import model as M


def table():
    out = {}
    for S in (64, 128, 256, 512, 1024):
        h = w = S
        rows = []
        for it in M.vgg_config('16'):
            if it == 'pool':
                h //= 2; w //= 2
                continue
            name, cin, cout = it
            if cin != 3:
                rows.append([name, h, w, cin, cout, M.conv_route(h, w, cin, cout), M.conv_route(h, w, cin, cout, dgrad=True)])
        out[str(S)] = rows
    return out


if __name__ == "__main__":
    assert not [k for k in os.environ if k.startswith("STROTSS_")], "the table is the DEFAULT policy: unset STROTSS_*"
    with open(os.path.join(ROOT, "tests", "golden", "route_table.json"), "w") as f:
        json.dump(table(), f, indent=0)
