"""Child process of tests/test_hip_regions.py: one rank of a masked (region-guided) step with the mask regions sharded
over the ranks (run_strotss.py:104-125 of the reference; nn/engine.py `dist_group`).  Launched with RANK / WORLD_SIZE /
MASTER_* in the environment (gloo on the box's single GPU); writes its losses, gradients and updated variables."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def problem(dev, world_group, h=96, w=128, regions=3, n_samples=256, deterministic=True):
    """Deterministic masked problem, identical in every process.
    deterministic=True: the tap adjoint as the sorted scatter.  Two processes share ONE GPU in this test, and under that
    contention the float-atomic scatter kernel was measured to lose updates in about one launch in ten (cache-line sized
    pieces of the sum, 1e-3..3e-2 of the gradient norm; never with the GPU to itself, never in the sorted scatter, never
    in stand-alone atomic probes -- DESIGN.md 6).  One process per GPU, the deployment, does not share."""
    from nn import _ops, engine, strotss_utils as SU
    from nn.model import VGGParams, synthetic_weights

    def img(hh, ww, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(1, hh, ww, 3, generator=g, dtype=torch.float32)
        return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1).contiguous()
    params = VGGParams(synthetic_weights('16', 0), '16', None, dev)
    content, style = img(h, w, 1).to(dev), img(h, w, 2).to(dev)
    cfeat = engine.extract_features(params, content)
    sfeat = engine.extract_features(params, style)
    rng = np.random.default_rng(0)
    edges = [round(r * w / regions) for r in range(regions + 1)]
    masks = []
    for a, b in zip(edges, edges[1:]):
        m = np.zeros((h, w), dtype=bool); m[:, a:b] = True
        masks.append(m)
    targets = []
    for m in masks:
        s_idx = torch.from_numpy(SU.make_indices_np(h, w, False, n_samples, rng, m)).to(dev)
        targets.append(engine.StyleTarget.build(_ops.hypercol_gather(sfeat, s_idx, False), int(s_idx.shape[0]), 2179))
    init = SU.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    alpha = 4.0
    eng = engine.StepEngine(params, cfeat, targets, init, alpha, 2.0 + alpha + 1.0 / alpha, 2e-3, sample_size=n_samples,
                            dist_group=world_group, deterministic=deterministic)
    idx = [[torch.from_numpy(SU.make_indices_np(h, w, True, n_samples, rng, m)).to(dev) for m in masks] for _ in range(3)]
    return eng, idx


def run(eng, idx, use_graph):
    if use_graph:
        eng.capture_graph(idx[0])
    out = {}
    eng.step(idx[0])
    torch.cuda.synchronize()
    out["losses0"] = eng.losses()
    out["gvars0"] = [g.cpu().clone() for g in eng.gvars]
    eng.step(idx[1])
    eng.step(idx[2])
    torch.cuda.synchronize()
    out["losses2"] = eng.losses()
    out["variables"] = [v.cpu().clone() for v in eng.variables]
    return out


if __name__ == "__main__":
    out_path, use_graph = sys.argv[1], sys.argv[2] == "graph"
    from nn import parallel
    torch.cuda.set_device(0)
    rank, world = parallel.init_from_env(0)
    assert world == int(os.environ["WORLD_SIZE"]) > 1
    dev = torch.device("cuda", 0)
    eng, idx = problem(dev, parallel.WORLD)
    assert eng.world == world and eng.my_regions == list(range(rank, eng.R, world))
    res = run(eng, idx, use_graph)
    res["my_regions"] = eng.my_regions
    res["two_graphs"] = eng._graph_post is not None
    torch.save(res, f"{out_path}.r{rank}.pt")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
