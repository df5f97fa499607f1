"""Child process of test_full_scale_winograd_vs_direct: ONE whole scale of the CLI's schedule (200 RMSprop steps at 128 px
through `run_strotss.run`, hipGraph replay, per-step host index draws from seed 0) under the convolution form the
environment selects (STROTSS_WINOGRAD is read once per process); writes the per-step losses and the output image."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

if __name__ == "__main__":
    out_path, content, style = sys.argv[1], sys.argv[2], sys.argv[3]
    lr = sys.argv[4] if len(sys.argv) > 4 else "2e-3"
    import run_strotss
    args = run_strotss.build_parser().parse_args([content, style, "-o", out_path + ".jpg", "--max_size", "128", "--start_level", "1",
                                                  "--level", "2", "--max_iter", "200", "--log_every", "200", "--lr", lr])
    trace = []
    final = run_strotss.run(args, trace=trace)
    rec = trace[0]
    losses = np.array([[s["loss"], s["loss_c"], s["loss_s"]] for s in rec["steps"]], np.float64)
    np.savez(out_path + ".npz", losses=losses, final=rec["final"].cpu().numpy(), u8=final.cpu().numpy())
