"""Child process of tests/test_hip_nccl_world1.py: ONE rank on ONE GPU over the `nccl` backend (= RCCL; a world of one is
legal RCCL).  STROTSS_DIST_FORCE=1 makes nn/parallel.py join the group and nn/engine.py keep the sharded structure, so
that what has only ever run over gloo executes on hardware at least once: `init_process_group("nccl", device_id=...)`,
the RCCL all-reduce of [pixel gradient | scalars] between the replays of two hipGraphs captured on a side stream, and the
device-identity all-gather of `ranks_share_a_gpu`.  The single device being selected is what nn/utils.py:73-85 of the
reference does with tf.config."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

if __name__ == "__main__":
    import torch.distributed as dist
    from nn import parallel
    import _region_worker as W
    assert os.environ["STROTSS_DIST_FORCE"] == "1" and os.environ.get("STROTSS_DIST_BACKEND", "nccl") == "nccl"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rank, world = parallel.init_from_env(0)
    out = {"rank": rank, "world": world, "backend": dist.get_backend(), "shares": parallel.ranks_share_a_gpu(None),
           "identity": parallel.device_identity()}
    # a bare RCCL all-reduce and a max-reduce (bench.py's aggregate_throughput) on device tensors
    t = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    want = t.clone()
    parallel.allreduce_sum_(t, parallel.WORLD)
    torch.cuda.synchronize()
    out["allreduce_identity"] = bool(torch.equal(t, want))
    v, el = parallel.aggregate_throughput(7.0, 2.0, parallel.WORLD, dev)
    out["aggregate"] = [v, el]
    # the masked step in its sharded form (one buffer, all-reduce before the fold adjoint) against the plain engine
    res = {}
    for name, group in (("plain", None), ("sharded", parallel.WORLD)):
        for mode in ("eager", "graph"):
            eng, idx = W.problem(dev, group)
            if group is not None:
                assert eng.sharded and eng.world == 1 and eng.my_regions == [0, 1, 2] and eng._reduce_buf is not None
            r = W.run(eng, idx, use_graph=(mode == "graph"))
            if group is not None and mode == "graph":
                assert eng._graph is not None and eng._graph_post is not None      # graph | RCCL all-reduce | graph
            res[(name, mode)] = r
    ref = res[("plain", "eager")]
    same = {}
    for key, r in res.items():
        same["/".join(key)] = bool(all(torch.equal(a, b) for a, b in zip(ref["variables"], r["variables"]))
                                   and all(torch.equal(a, b) for a, b in zip(ref["gvars0"], r["gvars0"]))
                                   and ref["losses0"] == r["losses0"] and ref["losses2"] == r["losses2"])
    out["bitwise_equal_to_plain_eager"] = same
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out))
