#!/usr/bin/env python3
"""bench.py -- optimisation steps/sec of the STROTSS inner loop on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (run_strotss.py:131-148: fold -> VGG16 -> hypercolumn
sampling -> self-similarity / moment / REMD / palette losses -> backward -> RMSprop) at the
1024x1024 scale of a synthetic 1024-px content/style pair with 1024 samples of D = 2179 -- the
configuration BASELINE.json's metric is quoted on.  Inputs (images, frozen weights, style
statistics, the index stream) are resident in HBM before the timed region; fp32 throughout (the
reference's dtype; the MFMA used is the exact-f32 one).  With N > 1 every rank optimises its own
pair (replicas, no collective on the data path) and `value` is the sum.  `--mode strips` instead shards
ONE pair over the N GPUs by image strips (nn/parallel.py; strong scaling, `value` = steps/s of that one job);
`--mode regions` is BASELINE config 4: ONE masked pair (4 mask regions), the regions dealt to the N ranks, one RCCL
all-reduce of the pixel gradient per step (strong scaling; the trunk is replicated).
A bare `python bench.py --gpus N` (no torchrun around it) starts the N ranks itself: it re-launches this script under
`python -m torch.distributed.run` as a CHILD process before anything touches the GPU and exits with the child's code.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      : the kernel class with the largest share of the step (at 1024 px: the fused F(4x4,3x3) Winograd kernel
                  on the f32 MFMA) against ITS bound, from the routing the library actually took per layer and direction
                  (strotss_conv3x3_winograd_route); `by_path` holds all three classes, each against its own bound:
                  f32-MFMA kernels vs 157.3 TFLOP/s, bf16x3 GEMMs vs 2500 / 6 TFLOP/s, Winograd transform kernels vs
                  8 TB/s; `blended` keeps rounds 1-2's all-conv-launches figure, labelled as a blend; `traffic` = HBM
                  bytes per launch of the dominant kernel from two `rocprofv3 --pmc` child passes of THIS run (falls back
                  to the committed profile, and says so, when rocprofv3 is missing);
  roofline_pairwise : the cosine cost-matrix GEMM (N x N x D) the metric names, same method;
  cpu_baseline  : the oracle (oracle/strotss_oracle.py, fp32 torch-CPU restatement of the
                  reference) timed on the host cores on a bounded sample of the same workload;
  pyramid       : steps/s at every scale 64..1024 and the projected optimisation wall-clock of the
                  full 5-scale x 200-step run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

F32_MFMA_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: dense f32 MFMA
BF16_MFMA_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA
SAMPLES = 1024
D = 2179


def synth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(1, h, w, 3, generator=g, dtype=torch.float32)
    return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1).contiguous()


def conv_flops(params, h, w):
    """algorithmic FLOP of one trunk forward (= one data-gradient pass) at (h, w)."""
    total, ch, cw = 0.0, h, w
    for it in params.cfg:
        if it == 'pool':
            ch, cw = ch // 2, cw // 2
        else:
            total += 2.0 * 9 * it[1] * it[2] * ch * cw
    return total


def region_masks(scale, regions):
    """`regions` vertical bands as boolean (scale, scale) masks (each >= 10000 px at 1024)."""
    if regions <= 1:
        return [None]
    edges = [round(r * scale / regions) for r in range(regions + 1)]
    out = []
    for a, b in zip(edges, edges[1:]):
        m = np.zeros((scale, scale), dtype=bool)
        m[:, a:b] = True
        out.append(m)
    return out


def build_engine(params, scale, dev, seed, sample_size=SAMPLES, strips=None, regions=1, dist_group=None):
    from nn import _ops, engine, strotss_utils as SU
    content = synth_image(scale, scale, 100 + seed).to(dev)
    style = synth_image(scale, scale, 200 + seed).to(dev)
    rng = np.random.default_rng(seed)
    cfeat = engine.extract_features(params, content)
    sfeat = engine.extract_features(params, style)
    targets = []
    for mk in region_masks(scale, regions):          # style samples per region (run_strotss.py:99-101 / 128)
        s_idx = torch.from_numpy(SU.make_indices_np(scale, scale, False, sample_size, rng, mk)).to(dev)
        feats = _ops.hypercol_gather(sfeat, s_idx, False)
        targets.append(engine.StyleTarget.build(feats, int(s_idx.shape[0]), D))
    del sfeat
    init = SU.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    alpha = 1.0            # alpha at the 5th scale of the schedule 16, 8, 4, 2, 1
    eng = engine.StepEngine(params, cfeat, targets, init, alpha, 2.0 + alpha + 1.0 / max(alpha, 1.0), 1e-3,
                            sample_size=sample_size, strips=strips, dist_group=dist_group)
    return eng, rng


def index_stream(scale, count, rng, dev, sample_size=SAMPLES, regions=1):
    """(count, regions, n, 2) index sets, uploaded before the timed region"""
    from nn import strotss_utils as SU
    masks = region_masks(scale, regions)
    arr = np.stack([np.stack([SU.make_indices_np(scale, scale, True, sample_size, rng, mk) for mk in masks])
                    for _ in range(count)])
    return torch.from_numpy(arr).to(dev)


def strip_index_stream(scale, count, rng, dev, plan, sample_size=SAMPLES, regions=1):
    """per step and region: the index set ordered by owning rank + its block offsets (identical on every rank: same seed).
    -> (list over steps of lists over regions of (n, 2) device tensors, list over steps of lists over regions of offsets)"""
    from nn import parallel as par, strotss_utils as SU
    masks = region_masks(scale, regions)
    idx, offs = [], []
    for _ in range(count):
        sets = [par.sort_indices_by_strip(SU.make_indices_np(scale, scale, True, sample_size, rng, mk), plan) for mk in masks]
        idx.append([torch.from_numpy(s[0]).to(dev) for s in sets])
        offs.append([s[1] for s in sets])
    return idx, offs


def run_steps(eng, idx, first, count):
    if eng._draw is not None:                 # the step draws its own index sets on the device (csrc/draw.hip)
        for _ in range(count):
            eng.step()
        return
    for i in range(first, first + count):
        eng.step(list(idx[i % idx.shape[0]]))


# reference device for `normalised`: the nominal figures of MI355X_MICROARCH.md (f32 MFMA 155 TF measured back to back,
# bf16 2.5 PF dense x the ~0.9 a register-only loop holds, copy 6.29 TB/s): a box that calibrates exactly there is left
# unchanged.  Weights = the shares of a 1024-px step by what bounds them (fused f32 Winograd kernel + first
# layer ~0.43, bf16x3 GEMMs ~0.32, transform / pooling / gather / fold kernels ~0.25).
CALIB_REF = {"mfma_f32": 155.0, "mfma_bf16": 2250.0, "copy": 5000.0}
CALIB_WEIGHTS = {"mfma_f32": 0.43, "mfma_bf16": 0.32, "copy": 0.25}


def normalise(steps_per_sec, calib):
    """steps/s this run would show on the reference device: every share of the step time is scaled by the box's measured
    rate for its bound (time_ref = time_box * sum_k w_k * box_k / ref_k)."""
    if not calib:
        return None
    rates = {"mfma_f32": calib["mfma_f32"]["tflops"], "mfma_bf16": calib["mfma_bf16"]["tflops"],
             "copy": calib["copy"]["GBps_read_plus_write"]}
    scale = sum(CALIB_WEIGHTS[k] * rates[k] / CALIB_REF[k] for k in rates)
    return {"steps_per_sec": round(steps_per_sec / scale, 2), "time_scale_box_over_reference": round(1.0 / scale, 4),
            "model": "time_ref = time_box * sum_k w_k * rate_box_k / rate_ref_k", "weights": CALIB_WEIGHTS, "reference": CALIB_REF,
            "validated": False,
            "note": "NOT a usable normalisation yet: over six boxes of one tree (profiles/r04_box_calibration_six_runs.txt) the raw "
                    "steps/s spread 4.1 % and this figure 5.3 %; neither the register-only loops, nor the loaded loop, nor the copy "
                    "rate track the boxes' step rate (one box calibrated slowest and stepped second fastest).  Compare runs by "
                    "long_window on the SAME box, alternating (tools/ab_env.sh); across boxes quote the range."}


def box_calibration(dev):
    """What THIS box sustains, measured right after the timed steps (warm chip): a ~45-ms register-only f32-MFMA loop and
    a ~45-ms bf16-MFMA loop (one wave per SIMD, every CU; rate and the clock the device held = s_memtime / s_memrealtime,
    median over workgroups) and a 256-MB streaming copy.  Boxes of the pool differ by up to 5 % in steps/s on one binary;
    these three numbers are what `normalised` is computed from."""
    from nn import _hip
    lib = _hip.lib()
    blocks = 256
    sink = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    clk = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
    st = _hip.stream_ptr()
    out = {}
    for name, bf16, iters, flop in (("mfma_f32", 0, 100000, 4096.0), ("mfma_bf16", 1, 200000, 32768.0)):
        _hip.check(lib.strotss_calib_mfma(bf16, blocks, 2000, sink.data_ptr(), clk.data_ptr(), st), "calib")     # code load
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _hip.check(lib.strotss_calib_mfma(bf16, blocks, iters, sink.data_ptr(), clk.data_ptr(), st), "calib")
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1)
            c = clk.cpu().numpy().reshape(blocks, 2).astype(np.float64)
            ghz = float(np.median(c[:, 0] / np.maximum(c[:, 1], 1.0)) * 0.1)
            tf = blocks * 4 * iters * 16.0 * flop / (ms * 1e-3) / 1e12
            if best is None or tf > best[0]:
                best = (tf, ghz, ms)
        out[name] = {"tflops": round(best[0], 1), "clock_ghz": round(best[1], 3), "ms": round(best[2], 2)}
    # the loaded loop (random operands re-read from LDS, two waves per SIMD): the one that tells boxes apart
    sink2 = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    _hip.check(lib.strotss_calib_mfma(2, blocks, 2000, sink2.data_ptr(), clk.data_ptr(), st), "calib")
    runs = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 60000
        e0.record()
        _hip.check(lib.strotss_calib_mfma(2, blocks, iters, sink2.data_ptr(), clk.data_ptr(), st), "calib")
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1)
        c = clk.cpu().numpy().reshape(blocks, 2).astype(np.float64)
        runs.append((blocks * 8 * iters * 4.0 * 32768.0 / (ms * 1e-3) / 1e12, float(np.median(c[:, 0] / np.maximum(c[:, 1], 1.0)) * 0.1), ms))
    runs.sort()
    out["mfma_bf16_lds_random"] = {"tflops": round(runs[2][0], 1), "clock_ghz": round(runs[2][1], 3), "ms": round(runs[2][2], 2),
                                   "tflops_min_max": [round(runs[0][0], 1), round(runs[-1][0], 1)]}
    # dependent-load latency in the L2 (2 MB chain), the Infinity Cache (64 MB) and HBM (1 GB): ns per hop, 64 chasers
    lat = {}
    g = torch.Generator().manual_seed(1)
    for name, n_el, steps in (("l2_2MB", 1 << 15, 20000), ("mall_64MB", 1 << 20, 10000), ("hbm_1GB", 1 << 24, 5000)):
        perm = torch.randperm(n_el, generator=g)
        nxt = torch.empty(n_el, dtype=torch.int64)
        nxt[perm] = perm.roll(-1)                                    # one cycle through all elements
        buf = torch.zeros(n_el * 16, dtype=torch.int32)
        buf[::16] = nxt.to(torch.int32)
        dbuf = buf.to(dev)
        sink3 = torch.zeros(64, dtype=torch.int32, device=dev)
        clk3 = torch.zeros(128, dtype=torch.int64, device=dev)
        for _ in range(2):
            _hip.check(lib.strotss_calib_chase(dbuf.data_ptr(), n_el // 64, 64, steps, sink3.data_ptr(), clk3.data_ptr(), st), "chase")
        torch.cuda.synchronize()
        c = clk3.cpu().numpy().reshape(64, 2).astype(np.float64)
        lat[name] = {"ns_per_hop": round(float(np.median(c[:, 1])) * 10.0 / steps, 1),
                     "cycles_per_hop": round(float(np.median(c[:, 0])) / steps, 1)}
        del dbuf
    out["latency"] = lat
    nbytes = 256 << 20
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    a.fill_(1)
    _hip.check(lib.strotss_calib_copy(a.data_ptr(), b.data_ptr(), nbytes, st), "calib_copy")
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.check(lib.strotss_calib_copy(a.data_ptr(), b.data_ptr(), nbytes, st), "calib_copy")
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    out["copy"] = {"GBps_read_plus_write": round(2.0 * nbytes / (best * 1e-3) / 1e9, 1), "ms": round(best, 4), "MB": nbytes >> 20}
    return out


FAMILIES = ["conv3x3_relu_fwd", "conv3x3_dgrad", "conv3x3_winograd_fwd", "conv3x3_winograd_dgrad", "step_losses_fwd_bwd", "index_draw", "selfsim_fwd_bwd",
            "remd_cos_fwd_bwd", "moment_fwd_bwd", "palette_remd_fwd_bwd", "hypercol_scatter", "maxpool2_bwd", "maxpool2_fwd",
            "conv3x3_c3_fwd", "conv3x3_c3_dgrad", "rmsprop_step", "resize_bilinear", "fold_pyramid", "resize_bilinear_adjoint",
            "loss_section"]


def time_kernel_families(eng, idx, steps=7, with_sites=False):
    """HIP-event time of every conv launch, of the loss entry points and of the HBM-bound families, on the stream the
    kernels are launched on (torch's current stream).  Eager launches (events cannot sit between the nodes of a replayed
    graph): one untimed eager step first, then `steps` timed ones; per call site (family, ordinal of the call inside the
    step) the MEDIAN over the steps is kept, so that a host hiccup between an event and its launch (in eager mode the GPU
    waits for the host on the small kernels; a Python GC pause there once put 43 ms on one scatter call) cannot leak into
    a family's figure.  The cyclic GC is off during the pass for the same reason.  (The pass that once reported a
    14.5 ms scatter -- BENCH_r01.json -- dropped the step's hipGraph right before its first eager step: the graph's
    destruction, hipGraphExecDestroy + the release of its capture pool, ran into that step.  The graph now stays
    alive until the pass is over.)"""
    import gc
    from nn import _ops, engine
    rec, step_no = [], [0]
    seen = {}

    def timed_call(name, orig):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # a ~40 us spin on the stream first: the host enqueues e0, the launches and e1 while the GPU is still busy, so
            # the interval holds the kernels' own time and not the host's launch latency (without it a 6 us resize
            # launch read 25 us: the idle GPU waiting for the next eager submission)
            torch.cuda._sleep(80000)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            key = (step_no[0], name)
            seen[key] = seen.get(key, -1) + 1
            rec.append((name, seen[key], step_no[0], e0, e1))
            return r
        return timed
    names = [n for n in FAMILIES if hasattr(_ops, n)]
    origs = {n: getattr(_ops, n) for n in names}
    gather_orig, losses_orig = engine.StepEngine._gather, engine.StepEngine._losses
    for n, o in origs.items():
        setattr(_ops, n, timed_call(n, o))
    engine.StepEngine._gather = timed_call("hypercol_gather", gather_orig)
    engine.StepEngine._losses = timed_call("loss_section", losses_orig)
    gc_was = gc.isenabled()
    gc.disable()
    try:
        for i in range(steps + 1):
            step_no[0] = i
            run_steps(eng, idx, i, 1)
            torch.cuda.synchronize()
    finally:
        if gc_was:
            gc.enable()
        for n, o in origs.items():
            setattr(_ops, n, o)
        engine.StepEngine._gather, engine.StepEngine._losses = gather_orig, losses_orig
    per_site = {}
    for name, ordinal, step, e0, e1 in rec:
        if step == 0:
            continue                          # the eager warm-up step
        per_site.setdefault((name, ordinal), []).append(e0.elapsed_time(e1))
    out, sites = {}, {}
    for (name, ordinal), ts in per_site.items():
        med = float(np.median(ts))
        sites[(name, ordinal)] = med
        t, c = out.get(name, (0.0, 0))
        out[name] = (t + med, c + 1)
    fam = {k: {"ms_per_step": v[0], "launches_per_step": v[1]} for k, v in out.items()}
    return (fam, sites) if with_sites else fam


ROUTE_NAMES = {0: "F(2x2,3x3): winograd_in_kernel -> gemm_kc_pipe_kernel x16 (f32 MFMA) -> winograd_out_kernel",
               1: "winograd43_fused_kernel (F(4x4,3x3), transforms + 36 products on chip, f32 MFMA)",
               2: "winograd43_in_kernel -> gemm_kc_pipe_kernel x36 (f32 MFMA) -> winograd43_out_kernel",
               3: "winograd43_in_x3_kernel -> gemm_x3_kernel<X3CfgK16<3>> x36 (bf16x3, 128x128 tiles) -> winograd43_out_kernel",
               4: "winograd43_in_x3_kernel -> gemm_x3_kernel<X3Cfg<64>> x36 (bf16x3, 64x64 tiles) -> winograd43_out_kernel",
               -1: "conv3x3_mfma_pipe_kernel / conv3x3_mfma_splitk_kernel (direct 3x3, f32 MFMA)"}


def conv_path_breakdown(eng, idx, params, sites7):
    """What the 12 generic conv layers of a step ACTUALLY ran (route per layer and direction from the library's own policy,
    strotss_conv3x3_winograd_route) and where their time went, in three classes, each against its own bound:
    f32-MFMA kernels (fused Winograd kernel, f32 GEMMs, direct conv) vs the dense f32 MFMA peak; bf16x3 GEMMs vs the dense
    bf16 peak / 6; transform kernels vs the HBM peak.  Three-kernel layers are split by timing the same eager steps with
    the library's stage mask at 1 (input transform only), 3 (+ GEMMs) and 7 (all): in = t1, GEMMs = t3 - t1, out = t7 - t3."""
    from nn import _hip, _ops
    lib = _hip.lib()
    tr = eng.trunk
    old = lib.strotss_debug_winograd_stages(3)
    try:
        _, sites3 = time_kernel_families(eng, idx, steps=5, with_sites=True)
        lib.strotss_debug_winograd_stages(1)
        _, sites1 = time_kernel_families(eng, idx, steps=5, with_sites=True)
    finally:
        lib.strotss_debug_winograd_stages(old)
    macs = {0: 9.0, 2: 4.0, 4: 2.25}
    cls = {"f32_mfma": {"gflop": 0.0, "ms": 0.0, "kernels": set()}, "bf16x3_gemm": {"gflop": 0.0, "ms": 0.0, "kernels": set()},
           "transforms": {"GB": 0.0, "ms": 0.0, "kernels": set()}}
    per_layer = []
    ordinals = {}
    plan = [st for st in tr.plan if st[0] == 'conv' and params.layers[st[1]]["cin"] != 3]
    for direction in ("fwd", "dgrad"):
        for st in (plan if direction == "fwd" else reversed(plan)):
            _, li, (kind, _si) = st
            L, a, t = params.layers[li], tr.acts[li], tr.wtile[li]
            h, w = int(a.shape[1]), int(a.shape[2])
            cin, cout = (L["cin"], L["cout"]) if direction == "fwd" else (L["cout"], L["cin"])
            name = {("fwd", True): "conv3x3_winograd_fwd", ("fwd", False): "conv3x3_relu_fwd",
                    ("dgrad", True): "conv3x3_winograd_dgrad", ("dgrad", False): "conv3x3_dgrad"}[(direction, t != 0)]
            site = (name, ordinals.get(name, 0))
            ordinals[name] = site[1] + 1
            t7 = sites7.get(site)
            if t7 is None:
                continue
            flop = 2.0 * macs[t] * cin * cout * h * w
            if t == 0:
                route = -1
            else:
                u = L["u_fwd" if direction == "fwd" else "u_bwd"][t]
                route = int(lib.strotss_conv3x3_winograd_route(h, w, cin, cout, t, int(_ops.winograd_packed(u) is not None),
                                                               int(_ops.winograd_x3(u, h, w) is not None)))
            rec = {"layer": L["name"], "dir": direction, "hw": [h, w], "cin": cin, "cout": cout, "route": route,
                   "ms": round(t7, 4)}
            if route in (-1, 1):
                cls["f32_mfma"]["gflop"] += flop / 1e9; cls["f32_mfma"]["ms"] += t7
                cls["f32_mfma"]["kernels"].add(ROUTE_NAMES[route])
            else:
                t1, t3 = sites1.get(site, 0.0), sites3.get(site, 0.0)
                t_in, t_gemm, t_out = t1, max(t3 - t1, 0.0), max(t7 - t3, 0.0)
                m = t + 2
                T = -(-h // t) * -(-w // t)
                x3 = route in (3, 4, 5, 6)
                v_bytes = m * m * T * cin * 4.0 * (1.5 if x3 else 1.0)
                m_bytes = m * m * T * cout * 4.0
                x_in, x_out = h * w * cin * 4.0, h * w * cout * 4.0
                has_mask = direction == "dgrad" and kind == "conv"
                tb = x_in + v_bytes                                     # input transform: reads X, writes V
                if route != 6:
                    tb += m_bytes + x_out * (2.0 if has_mask else 1.0)  # output transform: reads M (+ mask), writes X
                    cls["transforms"]["ms"] += t_in + t_out
                else:                                                   # output transform lives in the GEMM kernel
                    cls["transforms"]["ms"] += t_in
                    t_gemm = max(t7 - t1, 0.0)
                cls["transforms"]["GB"] += tb / 1e9
                cls["transforms"]["kernels"].add("winograd43_in_x3_kernel" if x3 else ("winograd43_in_kernel" if t == 4 else "winograd_in_kernel"))
                if route != 6:
                    cls["transforms"]["kernels"].add("winograd43_out_kernel" if t == 4 else "winograd_out_kernel")
                key = "bf16x3_gemm" if x3 else "f32_mfma"
                cls[key]["gflop"] += flop / 1e9; cls[key]["ms"] += t_gemm
                cls[key]["kernels"].add(ROUTE_NAMES[route])
                rec.update({"in_ms": round(t_in, 4), "gemm_ms": round(t_gemm, 4), "out_ms": round(t_out, 4)})
            per_layer.append(rec)
    out = {}
    for k, peak, unit in (("f32_mfma", F32_MFMA_PEAK_TFLOPS, "TFLOP/s"), ("bf16x3_gemm", BF16_MFMA_PEAK_TFLOPS / 6.0, "TFLOP/s")):
        c = cls[k]
        if c["ms"] > 0:
            ach = c["gflop"] / c["ms"]
            out[k] = {"kernels": sorted(c["kernels"]), "executed_gflop_per_step": round(c["gflop"], 1),
                      "ms_per_step": round(c["ms"], 3), "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit,
                      "frac": round(ach / peak, 4)}
    c = cls["transforms"]
    if c["ms"] > 0:
        rate = c["GB"] / (c["ms"] * 1e-3)
        out["transforms"] = {"kernels": sorted(c["kernels"]), "algorithmic_GB_per_step": round(c["GB"], 3),
                             "ms_per_step": round(c["ms"], 3), "achieved": round(rate, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(rate / HBM_PEAK_GBPS, 4), "bound": "hbm"}
    return out, per_layer


def pairwise_roofline(dev, iters=50):
    """The N x N x D cosine cost matrix (losses.py:12-15) on its own, as the loss entry points run it: the bf16x3
    GEMM core (csrc/mfma_x3.h: f32 operands split exactly into three bf16 planes, six exact partial products, f32
    accumulation) with the fused cosine epilogue.  Algorithmic 2*N*N*D f32 FLOP / HIP-event time of one launch,
    against the bound of the MFMA it issues: dense bf16 peak / 6 (the bf16 MFMA executes 6x the f32-equivalent FLOP)."""
    from nn import _ops
    g = torch.Generator().manual_seed(3)
    x = torch.zeros(SAMPLES, _ops.pad32(D)); x[:, :D] = torch.relu(torch.randn(SAMPLES, D, generator=g))
    y = torch.zeros(SAMPLES, _ops.pad32(D)); y[:, :D] = torch.relu(torch.randn(SAMPLES, D, generator=g))
    x, y = x.to(dev), y.to(dev)
    ld = int(x.shape[1])
    x3 = os.environ.get("STROTSS_X3", "1") != "0"
    if x3:
        rx, px = _ops.row_inv_norm_x3(x, SAMPLES)
        ry, py = _ops.row_inv_norm_x3(y, SAMPLES)
        fn = lambda: _ops.cosine_distance_x3(px, rx, SAMPLES, py, ry, SAMPLES, ld)
    else:
        rx, ry = _ops.row_inv_norm(x, SAMPLES), _ops.row_inv_norm(y, SAMPLES)
        fn = lambda: _ops.cosine_distance(x, rx, SAMPLES, y, ry, SAMPLES)
    for _ in range(5):
        fn()
    reps = []
    for _ in range(3):              # the median of three batches of `iters` launches (one batch was once seen 5x slow)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / iters)
    ms = sorted(reps)[1]
    tf = 2.0 * SAMPLES * SAMPLES * D / (ms * 1e-3) / 1e12
    kern = ("gemm_x3_kernel<X3Cfg<64>, EpiCosDistX3> (cosine cost matrix 1024x1024x2179, style x prediction: full matrix; "
            "bf16x3 core)" if x3 else "gemm_kc_pipe_kernel<64,64,EpiCosDist> (cosine cost matrix 1024x1024x2179, f32 MFMA)")
    # the bound of the instruction the kernel issues: the bf16x3 core spends 6 bf16 MFMA products per f32 product, so its
    # f32-equivalent ceiling is the dense bf16 peak / 6; with STROTSS_X3=0 the kernel runs on the f32 MFMA itself
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if x3 else F32_MFMA_PEAK_TFLOPS
    # what actually bounds it (measured in rounds 2-4): the bytes a CU can take in.  A 64 x 64 tile stages
    # (64 + 64 rows) x 64 B x 3 planes = 24 KiB per K-step of 32 through LDS-DMA; 256 tiles x 69 K-steps = 424 MB per launch, every
    # byte entering some CU's LDS.  The guide's measured ceiling for L2-resident rows gathered into LDS is 66-73 GB/s per CU
    # (MI355X_MICROARCH.md, "Indexed rows: gather into LDS"): 256 CUs x 70 GB/s = 17.9 TB/s.
    ingest = None
    if x3:
        tiles, ksteps = (SAMPLES // 64) ** 2, ld // 32
        gb = tiles * ksteps * (64 + 64) * 64 * 3 / 1e9
        ingest = {"bytes_into_lds_per_launch_MB": round(gb * 1e3, 1), "achieved_TBps": round(gb / (ms * 1e-3) / 1e3, 2),
                  "peak_TBps": 17.9, "frac": round(gb / (ms * 1e-3) / 1e3 / 17.9, 3),
                  "peak_is": "256 CUs x 70 GB/s: the guide's measured LDS-DMA ingest of L2-resident rows per CU"}
    return {"kernel": kern, "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(tf / peak, 4), "avg_launch_us": round(ms * 1e3, 2), "traffic": None, "cu_ingest": ingest,
            "peak_is": "dense bf16 MFMA 2500 TFLOP/s / 6 partial products (f32-equivalent)" if x3 else "dense f32 MFMA",
            "executed_bf16_tflops": round(6 * tf, 1) if x3 else None, "bf16_mfma_peak_tflops": BF16_MFMA_PEAK_TFLOPS if x3 else None,
            "vs_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4)}


HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def hbm_families(fam, params, S):
    """The HBM-bound families of a step against the HBM peak: ALGORITHMIC bytes (each tensor read / written once) over
    the family's HIP-event time ('each family against its own bound')."""
    pooled, ch, cw, first_out = 0.0, S, S, None
    for it in params.cfg:
        if it == 'pool':
            ch, cw = ch // 2, cw // 2
            pooled += ch * cw * prev_c
        else:
            prev_c = it[2]
            if first_out is None:
                first_out = S * S * it[2]
    img = 3.0 * S * S
    lv = [3.0 * max(S >> k, 1) ** 2 for k in range(6)]                # elements of the 6 Laplacian variables
    pyr = sum(lv)
    ld = (D + 31) // 32 * 32
    gb = {
        "maxpool2_bwd": pooled * (4 + 1 + 16),                         # pooled gradient + argmax code in, 4 gradients out
        "conv3x3_c3_fwd": 4.0 * (img + first_out),                     # image in, 64-channel map out
        "conv3x3_c3_dgrad": 4.0 * (first_out + img),                   # 64-channel gradient in, pixel gradient out
        "rmsprop_step": 4.0 * 5 * pyr,                                 # var, rms, grad in; var, rms out
        # fold: level k = v[k] + up(level k+1): read the coarser level + the variable, write the level (5 launches)
        "resize_bilinear": 4.0 * sum(lv[k + 1] + 2 * lv[k] for k in range(5)),
        # the same fold as ONE launch (strotss_fold_pyramid): every level read once, the image written once
        "fold_pyramid": 4.0 * (pyr + lv[0]),
        # fold adjoint: level k gradient = up^T(level k-1 gradient): read the finer, write the coarser (5 launches)
        "resize_bilinear_adjoint": 4.0 * sum(lv[k - 1] + lv[k] for k in range(1, 6)),
        # hypercolumn gather, content + prediction: 4 bilinear taps x N x D read, N x ld written, per call
        "hypercol_gather": 2 * 4.0 * (4 * SAMPLES * D + SAMPLES * ld),
        # scatter: N x D gradients read, 4 taps x N x D float atomics (memory-side read-modify-write: counted once)
        "hypercol_scatter": 4.0 * (SAMPLES * D + 4 * SAMPLES * D),
    }
    out = {}
    for name, nbytes in gb.items():
        if name in fam and fam[name]["ms_per_step"] > 0:
            rate = nbytes / (fam[name]["ms_per_step"] * 1e-3) / 1e9
            out[name] = {"algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(rate, 1),
                         "frac_of_hbm_peak": round(rate / HBM_PEAK_GBPS, 3)}
    return out


TRAFFIC_CSV = next((p for p in (os.path.join(ROOT, "profiles", f"r{r:02d}_hbm_traffic_by_kernel.csv") for r in range(9, 0, -1))
                    if os.path.exists(p)), os.path.join(ROOT, "profiles", "r01_hbm_traffic_by_kernel.csv"))


def measure_traffic_live(scale):
    """HBM traffic per kernel launch measured IN THIS RUN: this script started twice as a child under `rocprofv3 --pmc
    FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; the program directly after `--`), three eager steps each.
    FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads,
    MI355X_MICROARCH.md).  {kernel name: {launches, read, write}} in bytes per launch, or None when rocprofv3 is absent or
    a pass fails (the caller then falls back to the committed profile and says so)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("STROTSS_BENCH_LIVE_PMC", "1") == "0":
        return None

    def short(name):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name)
        return re.sub(r"\(.*\)$", "", name)
    acc = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                odir = os.path.join(tmp, counter)
                cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", odir, "--", sys.executable, os.path.abspath(__file__),
                       "--traffic-child", "--scale", str(scale)]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, timeout=400)
                files = glob.glob(os.path.join(odir, "**", "*_counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    print(f"bench.py: rocprofv3 --pmc {counter} pass failed (rc {r.returncode}); traffic from the committed profile",
                          file=sys.stderr)
                    return None
                with open(files[0]) as f:
                    for row in csv.DictReader(f):
                        if row["Counter_Name"] != counter:
                            continue
                        e = acc.setdefault(short(row["Kernel_Name"]), {"FETCH_SIZE": [], "WRITE_SIZE": []})
                        e[counter].append(float(row["Counter_Value"]))
    except Exception as exc:                                   # the measurement is optional; the bench line is not
        print(f"bench.py: live PMC pass failed ({exc!r}); traffic from the committed profile", file=sys.stderr)
        return None
    out = {}
    for k, e in acc.items():
        if e["FETCH_SIZE"] and e["WRITE_SIZE"]:
            out[k] = {"launches": len(e["FETCH_SIZE"]), "read": float(np.mean(e["FETCH_SIZE"])) * 1024 * 2,
                      "write": float(np.mean(e["WRITE_SIZE"])) * 1024}
    return out or None


def traffic_child(scale):
    """Child of measure_traffic_live (runs under rocprofv3 --pmc): one warm-up and three eager steps of the bench step."""
    from nn.model import VGGParams, synthetic_weights
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    params = VGGParams(synthetic_weights('16', 0), '16', None, dev)
    eng, rng = build_engine(params, scale, dev, seed=0)
    idx = index_stream(scale, 4, rng, dev)
    run_steps(eng, idx, 0, 4)
    torch.cuda.synchronize()


def pmc_traffic(path=TRAFFIC_CSV):
    """HBM bytes per conv launch from the committed PMC passes (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    in separate runs of this script, condensed by tools/summarize_rocprof.py with the gfx950 FETCH_SIZE x2
    correction).  One conv launch = one C-ABI call = input transform + batched GEMM + output transform, so the
    family's bytes are divided by its GEMM launches.  None when the summary is absent."""
    import csv
    if not os.path.exists(path):
        return None
    fam = ("winograd", "gemm_kc_pipe_kernel", "gemm_x3_kernel", "conv3x3_mfma_pipe_kernel")   # fused kernel: one launch
    total, launches = 0.0, 0
    with open(path) as f:
        rd = csv.reader(f)
        next(rd)
        for name, _grid, n, r_mb, w_mb in rd:
            if not name.startswith(fam) or (name.startswith("gemm_") and "EpiScaleStore" not in name):
                continue                                # GEMMs of the loss section carry other epilogues
            if r_mb == "nan" or w_mb == "nan":
                continue
            total += int(n) * (float(r_mb) + float(w_mb)) * 1e6
            if not name.startswith("winograd") or name.startswith("winograd43_fused"):
                launches += int(n)
    return total / launches if launches else None


def pmc_traffic_pairwise(path=TRAFFIC_CSV):
    """Fabric bytes (FETCH_SIZE x2 + WRITE_SIZE, same passes as pmc_traffic) of one full 1024 x 1024 cosine cost-matrix
    launch on the bf16x3 core (grid 256 workgroups x 256 threads = the style x prediction matrix).  Algorithmic: 2 x
    13.6 MB of x3 panels + 4.2 MB out; every XCD streams the whole B panel through its own L2."""
    import csv
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rd = csv.reader(f)
        next(rd)
        for name, grid, _n, r_mb, w_mb in rd:
            if name.startswith("gemm_x3_kernel<X3Cfg<64") and "EpiCosDistX3" in name and int(grid) == 65536:
                return (float(r_mb) + float(w_mb)) * 1e6
    return None


def wall_clock_to_output(dev, size=1024, level=5, max_iter=200):
    """The reference's Timer scope (run_strotss.py:44-45,159): model build + image load + all scales + postprocess
    + JPEG write, through the CLI's run() on a synthetic `size`-px pair written to a temporary directory."""
    import contextlib
    import logging
    import tempfile
    import run_strotss
    from nn import utils
    logging.getLogger('STROTSS').setLevel(logging.WARNING)       # stdout carries the JSON line only
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(sys.stderr):
        paths = []
        for name, seed in (("content.jpg", 100), ("style.jpg", 200)):
            paths.append(os.path.join(tmp, name))
            utils.write_image(synth_image(size, size, seed) * 255.0, paths[-1])
        out_path = os.path.join(tmp, "out.jpg")
        # the trunk's weights come from a file, as in the reference (model.py:31-33 loads its cached vgg16_norm.h5 inside
        # the timed scope): the seeded synthetic ones, written as the .npz the CLI's --weights takes, before the clock starts
        from nn import model as _model
        wpath = os.path.join(tmp, "vgg16_synthetic.npz")
        names = [it[0] for it in _model.vgg_config('16') if it != 'pool']
        np.savez(wpath, **{f"{n}/{k}": t.numpy() for n, (w, b) in zip(names, _model.synthetic_weights('16', 0))
                           for k, t in (("kernel", w), ("bias", b))})
        args = run_strotss.build_parser().parse_args(
            [paths[0], paths[1], "-o", out_path, "--max_size", str(size), "--level", str(level), "--max_iter",
             str(max_iter), "--log_every", str(max_iter), "--weights", wpath])
        # the CLI draws every step's index set inside the step, on the device (as the reference's traced step does,
        # strotss_utils.py:83-121)
        per_scale, orig = {}, run_strotss._optimise_scale

        def timed_scale(eng, scl, *a, **k):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r = orig(eng, scl, *a, **k)
            torch.cuda.synchronize()
            per_scale[str(scl)] = round(max_iter / (time.perf_counter() - t1), 1)
            return r
        run_strotss._optimise_scale = timed_scale
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_strotss.run(args)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            run_strotss._optimise_scale = orig
    return {"seconds": round(dt, 2), "config": f"{size}px synthetic pair, --level {level} --max_iter {max_iter} "
            f"(scales 64..{64 << (level - 1)}), incl. VGG build from a weight file (.npz, 59 MB), JPEG decode/encode, per-scale setup, "
            "hipGraph capture",
            "cli_steps_per_sec_by_scale": per_scale,
            "cli_steps_per_sec_is": "max_iter / wall time of the scale's loop INCLUDING its hipGraph capture (the per-step index "
                                    "draw is the first kernel of the graph)"}


def usable_cores():
    """Cores this process may actually run on: os.cpu_count() counts the whole host, the scheduler affinity mask and the
    cgroup CPU quota say what the container gets (a GPU box of this pool reports 256 and grants 16: 256 torch threads on
    16 cores ran the oracle 50x SLOWER than 16 threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(quota / int(g.read()) + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(scale, budget_s=25.0):
    """fp32 torch-CPU oracle (the restatement of the reference) on the same synthetic workload."""
    from oracle import strotss_oracle as O
    from nn.model import synthetic_weights
    host_cores = os.cpu_count() or 1
    threads = int(os.environ.get("STROTSS_CPU_THREADS", "0")) or usable_cores()   # all cores this process is granted
    torch.set_num_threads(threads)
    weights = synthetic_weights('16', 0)
    vgg = O.VGG(weights, dtype=torch.float32)
    content, style = synth_image(scale, scale, 100), synth_image(scale, scale, 200)
    rng = np.random.default_rng(0)
    with torch.no_grad():
        cf = [content] + vgg(content)
        sf = [style] + vgg(style)
        ss = O.sample_features(sf, O.make_indices(scale, scale, False, SAMPLES, rng), False)
    del sf
    init = O.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    variables = [v.clone().requires_grad_(True) for v in O.make_laplacian_pyramid(init)]
    rms = [torch.zeros_like(v) for v in variables]
    alpha, denom = 1.0, 4.0
    def one_step():
        idx = O.make_indices(scale, scale, True, SAMPLES, rng)
        t0 = time.perf_counter()
        res = O.train_step(variables, vgg, cf, ss, idx, alpha, denom)
        with torch.no_grad():
            for v, r, g in zip(variables, rms, res["grads"]):
                O.rmsprop_update(v, r, g, 1e-3)
        return time.perf_counter() - t0

    warm = one_step()
    if warm > budget_s / 2:            # one step already eats the budget: report it (no warm-up)
        timed, t_total, note = 1, warm, "no warm-up"
    else:
        timed, t_total, note = 0, 0.0, "after 1 warm-up"
        while timed < 5 and (timed == 0 or t_total + t_total / timed <= budget_s):
            t_total += one_step()
            timed += 1
    out = {"value": round(timed / t_total, 5), "unit": "steps/s", "cores": threads, "kind": "port",
           "host_cpu_count": host_cores, "usable_cores": usable_cores(),
           "sample": f"{timed} step(s) of the {scale}x{scale} scale (1024 samples) {note}, fp32 torch-CPU "
                     f"oracle with {threads} threads = every core this process is granted (os.cpu_count() = {host_cores} "
                     f"on the host, affinity mask / cgroup quota = {usable_cores()}; STROTSS_CPU_THREADS overrides) "
                     f"(the reference pins TF to 1 thread, nn/rand.py:16-17: see one_thread)"}
    # the reference's own setting (1 inter-op + 1 intra-op thread, nn/rand.py:16-17): ONE step of the same workload
    # when the multi-threaded rate says it fits the budget (SURVEY.md 8d asks for both)
    if os.environ.get("STROTSS_CPU_1THREAD", "1") != "0" and threads > 1 and threads / out["value"] <= 2.0 * budget_s:
        torch.set_num_threads(1)
        t1 = one_step()
        torch.set_num_threads(threads)
        out["one_thread"] = {"value": round(1.0 / t1, 5), "unit": "steps/s", "cores": 1,
                             "sample": f"1 step of the {scale}x{scale} scale, no warm-up, torch.set_num_threads(1)"}
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without torchrun around it: start the N ranks as a CHILD process (this process has
    not touched the GPU -- `import torch` does not -- and never will) and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=dict(os.environ))


def rehearse(args, world, rank):
    """`--rehearse`: the N-rank harness alone (rendezvous, barriers, max-over-ranks timing, rank 0's one JSON line) on
    a stand-in step that sleeps -- no kernel runs, nothing is measured; for the gloo world-size-2 CPU test of the
    launcher.  The line says so in `data`."""
    import torch.distributed as dist
    from nn import parallel
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    step_s = 0.01 * (1 + rank)                  # uneven ranks: the slowest sets the time
    for _ in range(args.warmup):
        time.sleep(step_s)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(step_s)
    if world > 1:
        dist.barrier()
    value, elapsed = parallel.aggregate_throughput(args.steps, time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "optimisation_steps_per_sec_1024px_pair", "value": round(value, 3), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "REHEARSAL: sleeping stand-in step, no GPU work",
                          "config": {"workload": "launcher rehearsal", "parallelism": f"replicas x{world} (gloo)"}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--host-draw", action="store_true", help="pre-drawn, pre-uploaded index sets (rounds 1-3) instead of the "
                                                             "draw kernel at the head of every timed step")
    ap.add_argument("--scale", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pyramid", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--no-e2e", action="store_true", help="skip the wall-clock-to-output run of the whole CLI schedule")
    ap.add_argument("--no-families", action="store_true", help="skip the per-kernel-family HIP-event pass")
    ap.add_argument("--halo", action="store_true", help="--mode strips: per-layer halo exchange instead of the recompute margin")
    ap.add_argument("--mode", choices=("replicas", "strips", "regions", "masked-strips"), default="replicas",
                    help="N > 1: independent pairs per GPU (default, weak scaling); ONE pair sharded by image strips; or ONE "
                         "masked pair (BASELINE config 4) with its mask regions dealt to the ranks and one all-reduce of the "
                         "pixel gradient per step; masked-strips: that masked pair sharded by image strips instead -- the sharding "
                         "that cuts trunk work (all strong scaling: value = steps/s of that one job)")
    ap.add_argument("--regions", type=int, default=4, help="--mode regions: number of mask regions (vertical bands)")
    ap.add_argument("--no-long-window", action="store_true", help="skip the extra >= 200-replay window and the box calibration (profiling passes)")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (traffic from the committed profile)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rehearse", action="store_true", help="N-rank harness only (gloo, sleeping stand-in step, no GPU)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))       # BEFORE any GPU call; a child, never an exec
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or without torchrun: bench.py starts the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        return rehearse(args, world, rank)
    if args.traffic_child:
        return traffic_child(args.scale)
    if world > 1:
        import torch.distributed as dist
        from nn import parallel as par
        local_rank %= max(1, torch.cuda.device_count())              # gloo rehearsal: N ranks on the box's one GPU
        par.init_from_env(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from nn.model import VGGParams, synthetic_weights
    from nn import parallel
    params = VGGParams(synthetic_weights('16', 0), '16', None, dev)
    S = args.scale
    strips, regions, group = None, 1, None
    if args.mode in ("strips", "masked-strips") and world > 1:
        strips = parallel.strip_plan(S, world, rank, halo=args.halo)
        if strips is None:
            raise SystemExit(f"--mode {args.mode}: sharding a {S}-row image over {world} ranks does not pay (see strip_plan)")
    if args.mode in ("regions", "masked-strips"):
        regions = args.regions
    if args.mode == "regions":
        group = parallel.WORLD if world > 1 else None
    one_job = strips is not None or args.mode in ("regions", "masked-strips")
    eng, rng = build_engine(params, S, dev, seed=0 if one_job else rank, strips=strips, regions=regions, dist_group=group)
    count = max(8, min(64, args.steps + args.warmup))
    if strips is not None:
        idx, offsets = strip_index_stream(S, count, rng, dev, strips, regions=regions)
        global run_steps
        run_steps = lambda e, ix, first, n: [e.step(ix[i % len(ix)], offsets[i % len(ix)]) for i in range(first, first + n)]
    else:
        idx = index_stream(S, count, rng, dev, regions=regions)
    # the step's index sets are drawn INSIDE the timed step, on the device (first node of the step's graph), as the
    # reference draws them inside its traced train_step (strotss_utils.py:83-121 via run_strotss.py:136); image strips
    # keep the host draw (every set is ordered by owning rank on the host)
    drawn = (strips is None and not args.host_draw
             and eng.enable_device_draw(0 if one_job else rank, 1000, region_masks(S, regions)))
    if not args.no_graph:
        if drawn:
            eng.capture_graph()
        else:
            eng.capture_graph(list(idx[0]), offsets[0] if strips is not None else None)

    run_steps(eng, idx, 0, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(eng, idx, args.warmup, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    value, elapsed = parallel.aggregate_throughput(args.steps, time.perf_counter() - t0, device=dev)
    if one_job:
        value /= world                    # ONE job: its steps are not multiplied by the ranks
    losses = eng.losses()
    # a window long enough to compare across runs whatever --steps was (the driver asks for 20 = 0.09 s): >= 200 replays
    long_window = None
    if world == 1 and not args.no_long_window:
        n_long = max(200, args.steps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(eng, idx, args.warmup + args.steps, n_long)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        long_window = {"steps": n_long, "ms_per_step": round(1e3 * el / n_long, 4), "steps_per_sec": round(n_long / el, 2)}
    calib = box_calibration(dev) if (rank == 0 and not args.no_long_window) else None

    out = None
    if rank == 0:
        n_gpus = world
        par_desc = (("one pair sharded by image strips (per-layer halo exchange with the neighbouring ranks), 2 all-reduces per step"
                     if strips.halo else "one pair sharded by image strips (halo recompute), 2 all-reduces per step")
                    + (f", {regions} mask regions (losses replicated, one all-reduce of all regions' feature rows)" if regions > 1 else "")
                    if strips is not None else
                    f"one masked pair, {regions} mask regions dealt round-robin to {n_gpus} rank(s), trunk replicated, "
                    f"1 all-reduce of the pixel gradient per step" if args.mode == "regions" else
                    "replicas (one pair per GPU)" if n_gpus > 1 else "single GPU")
        graph_mode = ("eager" if (args.no_graph or (strips is not None and eng._strip_graphs is None)) else
                      "3 hipGraphs per step around the two all-reduces" if strips is not None else
                      "2 hipGraphs per step around the all-reduce" if (world > 1 and args.mode == "regions") else "hipGraph")
        out = {"metric": "optimisation_steps_per_sec_1024px_pair", "value": round(value, 3), "unit": "steps/s",
               "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong" if one_job else "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{S}px content/style pair, {S}x{S} scale of the coarse-to-fine pyramid, "
                                      f"{SAMPLES} samples x D={D}, VGG16 (seeded He-normal weights), "
                                      f"RMSprop pixel update" + (f", {regions} mask regions" if regions > 1 else ""),
                          "scale_px": S, "samples": SAMPLES, "parallelism": par_desc},
               "loss_after": round(losses["loss"], 5), "launch_mode": graph_mode,
               "index_draw": ("inside the timed step: strotss_index_draw is the first kernel of every step's graph (Philox "
                              "stream, csrc/draw.hip; reference: strotss_utils.py:83-121 inside the traced step)" if drawn else
                              "outside the timed region: index sets drawn on the host and uploaded before the timed steps"),
               "long_window": long_window, "box_calibration": calib}
        out["normalised"] = normalise(long_window["steps_per_sec"] if long_window else value, calib)
        if world == 1 and strips is None and not args.no_families:
            # ---- per-kernel-family HIP-event timing (separate, untimed pass; eager launches, medians)
            saved_graph, eng._graph = eng._graph, None
            fam, sites = time_kernel_families(eng, idx, with_sites=True)
            bad = {k: round(v["ms_per_step"], 4) for k, v in fam.items() if v["ms_per_step"] >= out["ms_per_step"]}
            if bad:                                   # a family cannot take longer than the step: measure once more
                fam, sites = time_kernel_families(eng, idx, with_sites=True)
                bad = {k: round(v["ms_per_step"], 4) for k, v in fam.items() if v["ms_per_step"] >= out["ms_per_step"]}
            eng._graph = saved_graph
            out["kernel_families_valid"] = not bad
            conv_names = [n for n in ("conv3x3_relu_fwd", "conv3x3_dgrad", "conv3x3_winograd_fwd", "conv3x3_winograd_dgrad")
                          if n in fam]
            conv_ms = sum(fam[n]["ms_per_step"] for n in conv_names)
            conv_launches = sum(fam[n]["launches_per_step"] for n in conv_names)
            c3 = 2.0 * 9 * 3 * 64 * S * S
            algo = 2.0 * (conv_flops(params, S, S) - c3)       # direct-form FLOP of the 12 MFMA conv layers, fwd + dgrad
            macs_per_out = {0: 9.0, 2: 4.0, 4: 2.25}          # direct, F(2x2,3x3), F(4x4,3x3)
            executed = 2.0 * sum(2.0 * macs_per_out[t] * L["cin"] * L["cout"] * a.shape[1] * a.shape[2]
                                 for L, a, t in zip(params.layers, eng.trunk.acts, eng.trunk.wtile) if L["cin"] != 3)
            if not bad and conv_ms > 0:
                saved_graph, eng._graph = eng._graph, None
                by_path, per_layer = conv_path_breakdown(eng, idx, params, sites)
                eng._graph = saved_graph
                live = measure_traffic_live(S) if not args.no_live_pmc else None
                # the dominant kernel class of the step by time: its own bound, its own traffic
                dom = max((k for k in by_path if k != "transforms"), key=lambda k: by_path[k]["ms_per_step"])
                d = by_path[dom]
                dom_kernel = {"f32_mfma": "winograd43_fused_kernel", "bf16x3_gemm": "gemm_x3_kernel"}[dom]
                traffic, traffic_src = None, None
                if live is not None:
                    hit = [v for k, v in live.items() if k.startswith(dom_kernel) and (dom != "bf16x3_gemm" or "EpiScaleStore" in k)]
                    n = sum(v["launches"] for v in hit)
                    if n:
                        traffic = sum(v["launches"] * (v["read"] + v["write"]) for v in hit) / n
                        traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this script "
                                       "(3 eager steps each), FETCH_SIZE x2 (gfx950 half-count), bytes per launch of " + dom_kernel)
                if traffic is None:
                    traffic = pmc_traffic()
                    traffic_src = ("committed profile, not this run (rocprofv3 not available or its pass failed): HBM bytes per conv "
                                   "launch (" + os.path.relpath(TRAFFIC_CSV, ROOT) + ")")
                tf_exec = executed / (conv_ms * 1e-3) / 1e12
                tf_direct = algo / (conv_ms * 1e-3) / 1e12
                out["roofline"] = {
                    "kernel": "; ".join(d["kernels"]) + " -- the kernel class with the largest share of the step ("
                              + f"{d['ms_per_step']:.2f} of {out['ms_per_step']:.2f} ms), fwd + dgrad",
                    "bound": "mfma", "achieved": d["achieved"], "peak": d["peak"], "unit": "TFLOP/s", "frac": d["frac"],
                    "achieved_is": "FLOP the MFMA executes in this class (Winograd-domain products: 2.25 MACs per output for "
                                   "F(4x4,3x3), 4 for F(2x2,3x3), 9 direct; bf16x3 layers at their f32-equivalent FLOP against "
                                   "bf16 peak / 6) / summed HIP-event time of the class's launches (eager, medians; the classes' "
                                   "times are measured call by call and are not additive to ms_per_step, which is a graph replay)",
                    "traffic": traffic, "traffic_source": traffic_src,
                    "by_path": by_path,
                    "blended": {"note": "ALL conv launches of a step in one figure, f32-MFMA and bf16x3 layers mixed: a blend of "
                                        "two bounds, kept for continuity with rounds 1-2 (their `frac`), not a roofline",
                                "executed_tflops": round(tf_exec, 2), "vs_f32_mfma_peak": round(tf_exec / F32_MFMA_PEAK_TFLOPS, 4),
                                "conv_ms_per_step": round(conv_ms, 3), "launches_per_step": conv_launches,
                                "mfma_executed_gflop_per_step": round(executed / 1e9, 1),
                                "direct_form_gflop_per_step": round(algo / 1e9, 1),
                                "direct_form_equivalent_tflops": round(tf_direct, 2),
                                "effective_speedup_vs_direct_form": round(algo / executed, 3)},
                    "layers": per_layer}
                if live is not None:
                    out["hbm_traffic_by_kernel_MB_per_launch"] = {
                        k: {"launches": v["launches"], "read": round(v["read"] / 1e6, 1), "write": round(v["write"] / 1e6, 1)}
                        for k, v in sorted(live.items(), key=lambda kv: -kv[1]["launches"] * (kv[1]["read"] + kv[1]["write"]))[:14]}
            else:
                out["roofline"] = None
                out["kernel_families_rejected"] = bad
            out["roofline_pairwise"] = pairwise_roofline(dev)
            if out["roofline_pairwise"].get("executed_bf16_tflops"):
                lv = [v for k, v in (live or {}).items() if k.startswith("gemm_x3_kernel<X3Cfg<64") and "EpiCosDistX3" in k] \
                    if (not bad and conv_ms > 0) else []
                if lv:
                    # the step's three cost matrices: two symmetric launches (136 of 256 tiles) and the full style x prediction one
                    out["roofline_pairwise"]["traffic"] = max(v["read"] + v["write"] for v in lv)
                    out["roofline_pairwise"]["traffic_source"] = "measured in this run (rocprofv3 --pmc child passes): the full-matrix launch"
                else:
                    out["roofline_pairwise"]["traffic"] = pmc_traffic_pairwise()
                    out["roofline_pairwise"]["traffic_source"] = "committed profile, not this run (" + os.path.relpath(TRAFFIC_CSV, ROOT) + ")"
            out["kernel_families_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(fam.items())}
            out["kernel_families_launches_per_step"] = {k: v["launches_per_step"] for k, v in sorted(fam.items())}
            out["hbm_bound_families"] = hbm_families(fam, params, S)
    if rank == 0 and not args.no_pyramid and world == 1 and args.mode == "replicas":
        del eng
        torch.cuda.empty_cache()
        pyr, total = {}, 0.0
        for s in (64, 128, 256, 512):
            e, r = build_engine(params, s, dev, seed=0)
            ix = index_stream(s, 16, r, dev)
            e_drawn = not args.host_draw and e.enable_device_draw(0, 1000, None)
            if not args.no_graph:
                e.capture_graph(None if e_drawn else list(ix[0]))
            n = 200 if s <= 256 else 100
            run_steps(e, ix, 0, 10)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(e, ix, 10, n)
            torch.cuda.synchronize()
            sps = n / (time.perf_counter() - t1)
            pyr[str(s)] = round(sps, 2)
            total += 200.0 / sps
            del e
        if S == 1024:
            pyr["1024"] = round(out["value"], 2)
            total += 200.0 / out["value"]
            out["pyramid"] = {"steps_per_sec_by_scale": pyr,
                              "projected_optimisation_wall_clock_s_5x200": round(total, 2)}
    if rank == 0 and not args.no_e2e and world == 1 and S == 1024 and args.mode == "replicas":
        out["wall_clock_to_output"] = wall_clock_to_output(dev)
    if rank == 0 and not args.no_cpu_baseline and world == 1 and args.mode == "replicas":
        out["cpu_baseline"] = cpu_baseline(S)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
