"""Child process of tests/test_hip_strips.py: one rank of a strip-sharded step with per-layer HALO EXCHANGE
(nn/parallel.py HaloExchange).  Launched with RANK / WORLD_SIZE / MASTER_* in the environment (gloo on the
box's single GPU: rows are staged through the host); writes its losses, gradients and updated variables."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def problem(dev, plan, h=128, w=96, n_samples=256, group=None):
    """Deterministic problem, identical in every process.  The tap adjoint is the sorted scatter: two processes share one
    GPU in this test (see tests/_region_worker.py)."""
    from nn import _ops, engine, strotss_utils as SU
    from nn.model import VGGParams, synthetic_weights

    def img(hh, ww, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(1, hh, ww, 3, generator=g, dtype=torch.float32)
        return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1).contiguous()
    params = VGGParams(synthetic_weights('16', 0), '16', None, dev)
    content, style = img(h, w, 1).to(dev), img(h // 2, w, 2).to(dev)
    cfeat = engine.extract_features(params, content)
    sfeat = engine.extract_features(params, style)
    rng = np.random.default_rng(0)
    s_idx = torch.from_numpy(SU.make_indices_np(h // 2, w, False, n_samples, rng)).to(dev)
    target = engine.StyleTarget.build(_ops.hypercol_gather(sfeat, s_idx, False), int(s_idx.shape[0]), 2179)
    init = SU.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    alpha = 4.0
    eng = engine.StepEngine(params, cfeat, [target], init, alpha, 2.0 + alpha + 1.0 / alpha, 2e-3, sample_size=n_samples,
                            strips=plan, dist_group=group, deterministic=True)
    idx = [SU.make_indices_np(h, w, True, n_samples, rng) for _ in range(3)]
    return eng, idx


def run(eng, idx, plan):
    from nn import parallel
    out = {}
    for it in range(3):
        a, offs = idx[it], None
        if plan is not None:
            a, offs = parallel.sort_indices_by_strip(a, plan)
        eng.step([torch.from_numpy(a).to(eng.variables[0].device)], offs)
        if it == 0:
            torch.cuda.synchronize()
            out["losses0"] = eng.losses()
            out["gvars0"] = [g.cpu().clone() for g in eng.gvars]
            out["pf0"] = eng.pf[0].cpu().clone()
    torch.cuda.synchronize()
    out["losses2"] = eng.losses()
    out["variables"] = [v.cpu().clone() for v in eng.variables]
    return out


if __name__ == "__main__":
    out_path, h = sys.argv[1], int(sys.argv[2])
    from nn import parallel
    torch.cuda.set_device(0)
    rank, world = parallel.init_from_env(0)
    dev = torch.device("cuda", 0)
    plan = parallel.strip_plan(h, world, rank, halo=True)
    assert plan is not None and plan.halo
    eng, idx = problem(dev, plan, h=h, group=parallel.WORLD)
    res = run(eng, idx, plan)
    res["messages"] = eng._halo.messages
    res["window"] = (plan.win0, plan.win1, plan.own0, plan.own1)
    torch.save(res, f"{out_path}.r{rank}.pt")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
