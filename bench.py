#!/usr/bin/env python3
"""bench.py -- headline benchmark: Chamfer distance, 1M-vs-1M fp32 uniform-random clouds (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank (one process per GPU) computes
``chamfer_distance`` of its own independent (x, y) pair, both clouds already resident in HBM. Pairs shard across
ranks with no data-path collective (weak scaling); the per-step scalars are gathered once with RCCL at the end.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description): metric/value = whole-job query-points/s, plus
  roofline      dominant kernel k_search1_flat<float> (both directions in one launch): SURVEY 8d algorithmic bytes of the
                Chamfer row (24 B per query-point) / HIP-event launch time vs the 8 TB/s HBM peak; measured HBM traffic and the
                issue-side counters (VALU busy, TA busy, SALU issue fraction, active lanes) come from profiles/hbm_traffic.json, which
                profiles/postprocess.py derives from the round's rocprofv3 --pmc passes (stamped with the commit they were collected at:
                roofline.counters_commit);
  cpu_baseline  the reference's nanoflann path on this box's host cores (median of 3; search-only seconds beside it);
  parity        the GPU result of the timed pair against the reference's result on the SAME arrays (value + both index arrays);
  configs       (N = 1, behind the timed region; --no-configs skips it) every BASELINE config c1 .. c5 run through the same code as
                ``--config cN``: ms_per_step, value, roofline, parity against the reference and its cpu_baseline per entry.

Other BASELINE configs (parity-test cases): ``--config c1|c2|c3|c4|c5`` prints the same kind of line for Chamfer 10k fp64,
k_nearest_neighbors k=1 1M/1M, k=16 4M/4M, batched Hausdorff 256k pairs (32 pairs per GPU), Chamfer bunny-vs-1M f64.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 1_000_000
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=N_POINTS, help="points per cloud (default: the headline 1M)")
    ap.add_argument("--config", default="headline", choices=["headline", "c1", "c2", "c3", "c4", "c5", "gauss", "cluster", "outlier", "normals", "morton", "voxel", "sinkhorn"])
    ap.add_argument("--pairs", type=int, default=32, help="c4: pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline line at N = 1: do not run BASELINE configs c1-c5 behind the timed region (the `configs` object)")
    ap.add_argument("--no-kernel-events", action="store_true", help="--config lines: skip the extra steps that time the dominant kernel with HIP events (profiler runs count the operator's calls)")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_chamfer(x, y, kind, timing=None):
    """chamfer_distance as the reference computes it (point_cloud_utils/__init__.py:84-120) on the oracle's KNN; returns
    (value, corrs_x_to_y, corrs_y_to_x) and adds the kd-tree build / search seconds of both KNN calls to `timing`."""
    import numpy as np
    import oracle
    t1, t2 = {}, {}
    _, cxy = oracle.knn(x, y, 1, kind=kind, timing=t1)
    _, cyx = oracle.knn(y, x, 1, kind=kind, timing=t2)
    cxy, cyx = cxy[:, 0], cyx[:, 0]
    dxy = np.linalg.norm(x[cyx] - y, axis=-1).mean()
    dyx = np.linalg.norm(y[cxy] - x, axis=-1).mean()
    if timing is not None:
        timing["build_s"] = t1.get("build_s", 0.0) + t2.get("build_s", 0.0)
        timing["search_s"] = t1.get("search_s", 0.0) + t2.get("search_s", 0.0)
    return dxy + dyx, cxy, cyx


def cpu_baseline(x, y, samples=3):
    """The reference's CPU path (oracle/_ref = its own nanoflann.hpp; else the C restatement) timed as the reference
    behaves: input copies + 3x kd-tree build + OpenMP search on all host cores, both directions, plus the numpy tail of
    chamfer_distance. Median of `samples` full 1M-vs-1M Chamfer evaluations (~2-20 s each). Returns (report, last result)."""
    import numpy as np
    import oracle
    oracle.build()
    kind = "ref" if oracle.have_ref() else "port"
    ts, searches, res = [], [], None
    for _ in range(samples):
        tm = {}
        t0 = time.perf_counter()
        res = cpu_chamfer(x, y, kind, tm)
        ts.append(time.perf_counter() - t0)
        searches.append(tm.get("search_s", 0.0))
    dt = float(np.median(ts))
    cores = os.cpu_count() or 1
    rep = {"value": (x.shape[0] + y.shape[0]) / dt, "unit": "query-points/s", "cores": cores if kind == "ref" else 1,
           "kind": "reference" if kind == "ref" else "port", "cpu_model": cpu_model(),
           "seconds": [round(t, 3) for t in ts], "search_only_seconds": [round(t, 3) for t in searches],
           "search_only_value": (x.shape[0] + y.shape[0]) / float(np.median(searches)) if min(searches) > 0 else None,
           "sample": f"median of {samples} full chamfer_distance {x.shape[0]}-vs-{y.shape[0]} fp32 evaluations of the GPU step's own pair "
                     f"(3x kd-tree build + search per direction, as the reference does; {dt:.2f} s)"}
    return rep, res


def cpu_obj(units, seconds, kind, sample, cores=None, unit="query-points/s"):
    """`cpu_baseline` object of a --config line: the reference's CPU path (kind "reference": oracle/_ref = its own nanoflann.hpp) or the
    oracle's restatement (kind "port") timed on this box's host cores on a stated sample of the same workload."""
    return {"value": units / seconds if seconds > 0 else None, "unit": unit, "cores": cores if cores is not None else (os.cpu_count() or 1),
            "kind": kind, "cpu_model": cpu_model(), "seconds": round(seconds, 3), "sample": sample}


def counters():
    """Issue-side counters and HBM traffic of the dominant kernel, from the round's rocprofv3 --pmc passes (tracked file)."""
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(tp))
    except Exception:
        return {}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import point_cloud_utils_amd as pcu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    # PCU_BENCH_SHARE_GPU=1: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (tests/test_gpu_multirank.py: two ranks on the
    # one GPU of the test box). Ranks then share devices and the collectives -- scalars only -- go through gloo on host tensors; RCCL
    # refuses two ranks on one device. The line says so in config.collectives; it is never what the driver measures.
    share = os.environ.get("PCU_BENCH_SHARE_GPU", "0") not in ("", "0")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = torch.device("cpu") if share else dev
    distributed = world > 1
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    args.coll_dev, args.share = coll_dev, share
    if args.config != "headline":
        return other_config(args, pcu, np, torch, dist, dev, rank, world, distributed, sync_all)

    n = args.points
    # pair p uses seeds 1000+2p / 1001+2p; one pair per rank
    x_h = np.random.default_rng(1000 + 2 * rank).random((n, 3), dtype=np.float32)
    y_h = np.random.default_rng(1001 + 2 * rank).random((n, 3), dtype=np.float32)
    x, y = torch.from_numpy(x_h).to(dev), torch.from_numpy(y_h).to(dev)
    results = [0.0] * max(args.steps, 1)          # per-step scalars stay on the host until the final gather

    pcu.set_timing(0)
    pcu.chamfer_distance(x, y)        # initialisation, not a step: creates the context, loads the code object, sizes the workspace
    for _ in range(args.warmup):
        pcu.chamfer_distance(x, y)
    # Roofline input: HIP events around the main search launch (k_search1_flat<float>, both directions), recorded by the
    # library on its launch stream INSIDE the timed region -- on every 8th step only, because each event is a ~6 us
    # bubble between kernels.
    KEV_EVERY = 8
    k_ms, k_n = 0.0, 0
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        timed = s % KEV_EVERY == 0
        if timed:
            pcu.set_timing(1)
        results[s] = float(pcu.chamfer_distance(x, y))
        if timed:
            st = pcu.last_stats()
            k_ms += st["ms_kernel_search"]; k_n += st["n_kernel_search"]
            pcu.set_timing(0)
    if distributed:   # the only collective of the job: gather the per-pair scalars (K floats per rank)
        res_t = torch.tensor(results, dtype=torch.float32, device=coll_dev)
        gathered = [torch.empty_like(res_t) for _ in range(world)]
        dist.all_gather(gathered, res_t)
    sync_all()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # phase breakdown (diagnostic, OUTSIDE the timed region: 3 extra steps with phase events on)
    idx_ms = tot_ms = srch_ms = 0.0
    if rank == 0:
        pcu.set_timing(2)
        for _ in range(3):
            pcu.chamfer_distance(x, y)
            st = pcu.last_stats()
            idx_ms += st["ms_index"] / 3; srch_ms += st["ms_search"] / 3; tot_ms += st["ms_total"] / 3
        pcu.set_timing(0)

    # protocol B of SURVEY 8d (diagnostic, OUTSIDE the timed region): end to end numpy -> numpy, i.e. incl. the H2D copies of both
    # clouds over PCIe and the Python shim; median of 5. Never the reported `value`.
    e2e_ms = None
    if rank == 0:
        ts = []
        for _ in range(6):
            t1 = time.perf_counter(); pcu.chamfer_distance(x_h, y_h); ts.append(time.perf_counter() - t1)
        e2e_ms = float(np.median(ts[1:])) * 1e3
    if rank == 0:
        steps = max(args.steps, 1)
        qpts_per_step = 2 * n * world
        value = qpts_per_step * steps / dt
        # Dominant kernel: k_search1_flat<float>, ONE launch for both directions = 2n queries, each against the other cloud's n
        # points. Algorithmic bytes of the Chamfer row (p = 2, no indices): every input read once per role,
        # 2 * 3 * 4 * (N + M) = 24 B per query-point -> 48 MB per launch at 1M-vs-1M (the fused epilogue writes no rows).
        alg_bytes = 24.0 * 2 * n
        avg_ms = k_ms / max(k_n, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        cnt = counters()
        traffic = cnt.get("k_search1_flat_f32_bytes_per_launch")
        roof = {"bound": "hbm", "actual_bound": "vmem_issue",
                "bound_note": "`bound` is the roofline BASELINE.json designates (achieved = SURVEY 8d algorithmic bytes / launch time against the HBM peak); what the "
                              "kernel actually runs into is the CU's texture-address path -- ~20 cycles per vector-memory (gather) instruction whatever its width, "
                              "vmem_insts_per_wave of them, ta_busy_frac -- with vector-ALU issue next (valu_busy_frac); DESIGN.md 4.2. It cannot be HBM-bound: "
                              "SURVEY 8d's consistency note",
                "kernel": "k_search1_flat<float> (both directions in one launch, fused Chamfer epilogue)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "measured_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                "alg_bytes_per_launch": alg_bytes, "alg_bytes_rule": "SURVEY 8d Chamfer row: 24 B per query-point",
                "knn_rows_bytes_per_launch": 36.0 * 2 * n,
                "knn_rows_note": "B_knn (36 B/query incl. the (d, idx) rows) applies to k_nearest_neighbors / return_index calls, not to this metric",
                "avg_launch_ms": avg_ms, "launches_timed": k_n,
                "timing": f"HIP events around the main search launch of every {KEV_EVERY}th timed step",
                "whole_op": {"alg_bytes_per_step": 48.0 * n * world, "achieved_GBps": 48.0 * n * world / (dt / steps) / 1e9,
                             "frac": 48.0 * n * world / (dt / steps) / 1e9 / (HBM_PEAK_GBS * world),
                             "note": "all launches of the step + host; SURVEY 8d: 2*3*4*(N+M) bytes"}}
        # what actually bounds the kernel (it is not HBM-bound and cannot be): VALU issue, with the vector-memory address path next
        for k in ("valu_busy_frac", "ta_busy_frac", "salu_issue_frac", "active_lane_frac", "valu_insts_per_wave", "salu_insts_per_wave",
                  "vmem_insts_per_wave", "valu_issue_frac_at_2_cycles", "counters_source"):
            if k in cnt:
                roof[k] = cnt[k]
        # `traffic` and the counter fields are NOT measured by this run: they are the rocprofv3 --pmc collection tracked in profiles/hbm_traffic.json,
        # stamped with the commit it was collected at (profiles/collect.sh); the timing fields above are live
        roof["counters_commit"] = cnt.get("commit")
        roof["counters_note"] = "traffic + issue counters: tracked rocprofv3 --pmc collection (profiles/hbm_traffic.json) taken at `counters_commit`; everything else in this object is measured by this run"
        out = {
            "metric": "query-points/s, Chamfer 1M-vs-1M fp32", "value": value, "unit": "query-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"chamfer_distance, {n}-vs-{n} fp32 U[0,1)^3 clouds, one independent pair per GPU per step, "
                                   "inputs resident in HBM, scalar results gathered once (RCCL all_gather)",
                       "points_per_cloud": n, "pairs_per_step": world, "parallelism": f"pairs x{world}",
                       **({"collectives": "gloo on host tensors: REHEARSAL, ranks share GPUs (PCU_BENCH_SHARE_GPU)"} if share else {})},
            "roofline": roof,
            "device_ms_per_step": {"index_build": idx_ms, "search": srch_ms, "total": tot_ms,
                                   "note": "3 extra steps outside the timed region, phase events on"},
            "chamfer": float(results[0]),
            "end_to_end_numpy": {"ms_per_call": e2e_ms, "value": (2 * n / (e2e_ms * 1e-3)) if e2e_ms else None, "unit": "query-points/s",
                                 "note": "SURVEY 8d protocol B: host arrays in, Python float out (24 MB H2D inside the call); median of 5, outside the timed region"},
        }
        ref = None
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0, N = 1 only
            out["cpu_baseline"], ref = cpu_baseline(x_h, y_h)
        if not args.no_parity:
            # parity on the arrays that were timed (N > 1: rank 0's pair): the reference's value and BOTH index arrays
            import oracle
            if ref is None:
                oracle.build()
                ref = cpu_chamfer(x_h, y_h, "ref" if oracle.have_ref() else "port")
            ch0, cxy0, cyx0 = ref
            _, cxy, cyx = pcu.chamfer_distance(x, y, return_index=True)
            rel = abs(float(results[0]) - float(ch0)) / abs(float(ch0))
            idx_equal = bool(np.array_equal(cxy.cpu().numpy(), cxy0) and np.array_equal(cyx.cpu().numpy(), cyx0))
            same_every_step = bool(all(r == results[0] for r in results[:steps]))
            out["parity"] = {"chamfer_rel": rel, "tol": 1e-4, "idx_equal": idx_equal, "reference_value": float(ch0),
                             "identical_over_steps": same_every_step,
                             "checker": "oracle/_ref (the reference's nanoflann.hpp)" if oracle.have_ref() else "oracle port"}
            assert rel <= 1e-4, f"Chamfer value differs from the reference: {results[0]} vs {ch0}"
            assert idx_equal, "Chamfer correspondences differ from the reference"
        # Every BASELINE config in front of the driver (N = 1 only; OUTSIDE the timed region above): c1-c5 through the same code as
        # `--config cN`, a few steps each, parity against the reference and its CPU time inside each entry.
        if world == 1 and not args.no_configs:
            del x, y
            out["configs"] = {}
            for cfg, steps_c, warm_c in (("c1", 50, 5), ("c2", 30, 3), ("c3", 8, 2), ("c4", 8, 2), ("c5", 30, 3)):
                sub = argparse.Namespace(**vars(args))
                sub.config, sub.steps, sub.warmup = cfg, steps_c, warm_c
                t_c = time.perf_counter()
                line = other_config(sub, pcu, np, torch, dist, dev, rank, world, distributed, sync_all, emit=False)
                line["wall_s"] = round(time.perf_counter() - t_c, 1)
                out["configs"][cfg] = line
                torch.cuda.empty_cache()
            # compact, LAST in the line (a stored tail of the output shows all five): ms per step + whether every parity flag of the entry is true
            def _ok(par):
                flags = [v for k_, v in par.items() if isinstance(v, bool)]
                tol_ok = all(par[k_] <= par.get("tol", 0) for k_ in ("chamfer_rel",) if k_ in par)
                return bool(flags) and all(flags) and tol_ok
            out["configs_summary"] = {c_: {"ms": round(l_["ms_per_step"], 4), "parity_ok": _ok(l_.get("parity", {}))} for c_, l_ in out["configs"].items()}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def config_roofline(cfg, alg, step_s, k_ms, k_calls):
    """Per-config roofline object. Dominant kernel of the operator = the main lane-per-query search launch (k_search1_flat for k = 1,
    k_search<K> otherwise), which processes every query of the call: its algorithmic bytes are the operator's, its duration is
    measured live (HIP events, see the timed loop) where the library brackets it, and cross-checked / replaced by the rocprofv3 average of
    the same command in profiles/config_kernels.json (written by profiles/postprocess.py from the round's collection), which also gives
    the measured HBM traffic (calibrated FETCH_SIZE + WRITE_SIZE) of that kernel and of the whole call."""
    try:
        ck = json.load(open(os.path.join(ROOT, "profiles", "config_kernels.json"))).get(cfg, {})
    except Exception:
        ck = {}
    live_ms = k_ms / k_calls if k_calls else None
    prof_ms = ck["dominant_avg_us"] * ck.get("dominant_launches_per_call", 1.0) / 1e3 if ck.get("dominant_avg_us") else None
    achieved = alg / step_s / 1e9
    roof = {"bound": "hbm", "actual_bound": ck.get("actual_bound", "vmem_issue / valu_issue (k = 1: gathers; k > 1: sorted insertion; DESIGN.md 4.2)"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": ck.get("hbm_bytes_per_call"), "counters_commit": ck.get("commit"), "alg_bytes_per_call": alg,
            "note": "whole operator call (all launches + host) against SURVEY 8d's algorithmic bytes; traffic = measured HBM bytes of the whole call "
                    "(rocprofv3 --pmc, profiles/config_kernels.json: a tracked collection, see `source`)",
            "source": ck.get("source"), "gpu_us_per_call_rocprof": ck.get("gpu_us_per_call")}
    # Two objects, each naming ONE kernel with ITS duration (round 5 mixed them: the collection's dominant kernel can be a build kernel -- c1, c5 --
    # while the library's HIP events bracket the search launch):
    #   dominant_kernel  the launch with the most GPU time per call in the tracked rocprofv3 collection: its name, its rocprofv3 duration, its traffic;
    #                    the live HIP-event duration is attached only when that kernel IS the bracketed search launch
    #   search_kernel    the call's main search launch, bracketed live by the library (HIP events on its launch stream, this build)
    dom_name = ck.get("dominant") or ""
    dom_is_search = dom_name.startswith("k_search")
    dom = {"kernel": ck.get("dominant"), "ms_rocprof": prof_ms, "traffic": ck.get("dominant_hbm_bytes_per_launch"),
           "ms_live_hip_events": live_ms if dom_is_search else None}
    use_ms = (live_ms if dom_is_search else None) or prof_ms
    if use_ms:
        dom["achieved"] = alg / (use_ms * 1e-3) / 1e9
        dom["frac"] = dom["achieved"] / HBM_PEAK_GBS
        dom["timing"] = ("HIP events around this launch, 4 extra steps outside the timed region (this build)" if (live_ms and dom_is_search)
                         else "rocprofv3 average of the same command in the tracked collection at `counters_commit` (may belong to an earlier build)")
        dom["note"] = "the operator's algorithmic bytes over this kernel's own duration"
    if live_ms:
        roof["search_kernel"] = {"kernel": "the call's main search launch(es) as bracketed by the library: k_search1_flat<T> (k = 1), k_search_runs<T,K> / k_search<T,K> (k > 1)",
                                 "ms_live_hip_events": live_ms, "achieved": alg / (live_ms * 1e-3) / 1e9, "frac": alg / (live_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "timing": "HIP events, 4 extra steps outside the timed region (this build)"}
    roof["dominant_kernel"] = dom
    return roof


def other_config(args, pcu, np, torch, dist, dev, rank, world, distributed, sync_all, emit=True):
    """BASELINE configs 1-5 (and the uneven-cloud / next-row workloads): same timing protocol (device-resident inputs, barrier + sync, max over
    ranks), parity against the oracle outside the timed region. One JSON line on rank 0 -- or, with emit=False (the headline run attaches
    every BASELINE config to its own line), the line's dict is returned instead (rank 0; no process-group teardown)."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import cloud
    oracle.build()
    kind = "ref" if oracle.have_ref() else "port"
    refkind = "reference" if kind == "ref" else "port"
    cores_knn = (os.cpu_count() or 1) if kind == "ref" else 1        # (the reference's OpenMP search; the C restatement is serial)
    cfg = args.config
    parity = {}
    cpu = {}                        # "obj": the line's cpu_baseline (filled by check(), which times the CPU path it compares with)
    if cfg in ("c2", "c3"):
        n, k = (1_000_000, 1) if cfg == "c2" else (4_000_000, 16)
        q, r = cloud(1000 + 2 * rank, n, np.float32), cloud(1001 + 2 * rank, n, np.float32)
        tq, tr = torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev)
        step = lambda: pcu.k_nearest_neighbors(tq, tr, k)
        units, s = n, 4
        alg = 3 * s * n + 3 * s * n + n * k * (s + 8)
        name = f"k_nearest_neighbors k={k}, {n}-vs-{n} fp32"
        def check():
            d, c = pcu.k_nearest_neighbors(tq, tr, k)
            t0 = time.perf_counter(); d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind=kind); t_cpu = time.perf_counter() - t0
            cpu["obj"] = cpu_obj(n, t_cpu, refkind, f"one full k_nearest_neighbors k={k} {n}-vs-{n} fp32 call on the GPU step's own arrays (input copies, 3x kd-tree build, "
                                 f"OpenMP search on all cores, as the reference does; {t_cpu:.2f} s)", cores=cores_knn)
            return {"idx_equal": bool(np.array_equal(c.cpu().numpy(), c0)), "dist_bits_equal": bool(np.array_equal(d.cpu().numpy(), d0)), "stats": pcu.last_stats()}
    elif cfg == "c1":          # BASELINE config 1: the reference's CPU path on two 10k fp64 clouds is the expected result; the GPU call is timed
        n = 10_000
        x, y = cloud(1000 + 2 * rank, n, np.float64), cloud(1001 + 2 * rank, n, np.float64)
        tx, ty = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        step = lambda: pcu.chamfer_distance(tx, ty)
        units, alg = 2 * n, 2 * 3 * 8 * 2 * n
        name = f"chamfer_distance, {n}-vs-{n} fp64 (reference CPU path = expected value)"
        def check():
            t0 = time.perf_counter(); ch0, cxy0, cyx0 = oracle.chamfer_distance(x, y, return_index=True, kind=kind); t_cpu = time.perf_counter() - t0
            ch, cxy, cyx = pcu.chamfer_distance(tx, ty, return_index=True)
            cpu["obj"] = cpu_obj(2 * n, t_cpu, refkind, f"one chamfer_distance(return_index=True) {n}-vs-{n} fp64 call, the config's own pair (below 100 000 rows the reference "
                                 f"searches on one thread, src/point_cloud_distance.cpp:29-30; {t_cpu:.3f} s)", cores=1)
            return {"idx_equal": bool(np.array_equal(cxy.cpu().numpy(), cxy0) and np.array_equal(cyx.cpu().numpy(), cyx0)),
                    "chamfer_rel": abs(float(ch) - float(ch0)) / float(ch0), "tol": 1e-6}
    elif cfg == "c4":
        # 32 pairs per GPU; pair p of the job's npairs x world pairs belongs to rank p mod world (batched.shard_pairs) and goes through
        # batched_hausdorff: the library's batch entry point per rank + the job's ONE all_gather of the scalars, inside the timed region
        npairs, n = args.pairs, 262144
        total = npairs * world
        from point_cloud_utils_amd import batched
        pairs = {p: (torch.from_numpy(cloud(1000 + 2 * p, n, np.float32)).to(dev), torch.from_numpy(cloud(1001 + 2 * p, n, np.float32)).to(dev))
                 for p in batched.shard_pairs(total, rank, world)}
        last = {}
        def step():
            last["rows"] = batched.batched_hausdorff(lambda p: pairs[p], total)      # (collective inside: every rank calls it)
            return last["rows"]
        units, alg = npairs * 2 * n, npairs * 2 * 3 * 4 * 2 * n
        name = f"hausdorff_distance, {npairs} independent {n}-vs-{n} fp32 pairs per GPU (batched_hausdorff: batch entry point + one all_gather)"
        def check():
            if "rows" not in last:
                return {"pairs_checked": 0, "note": "no step ran"}
            rows = last["rows"]               # rank 0 only: the rows the last timed step gathered (no further collective here)
            ok = True
            t_cpu = 0.0
            for p in sorted(pairs):          # every pair of this rank against the reference
                a_, b_ = pairs[p][0].cpu().numpy(), pairs[p][1].cpu().numpy()
                t0 = time.perf_counter(); h0 = oracle.hausdorff_distance(a_, b_, return_index=True, kind=kind); t_cpu += time.perf_counter() - t0
                ok &= tuple(rows[p]) == tuple(float(v) for v in h0)
            cpu["obj"] = cpu_obj(len(pairs) * 2 * n, t_cpu, refkind, f"hausdorff_distance of this rank's {len(pairs)} pairs ({n}-vs-{n} fp32), one after the other: the reference "
                                 f"searches Hausdorff on ONE thread (num_threads = 0, src/point_cloud_distance.cpp:219); {t_cpu:.1f} s", cores=1)
            return {"pairs_checked": len(pairs), "tuples_equal": bool(ok)}
    elif cfg in ("gauss", "cluster", "outlier"):      # uneven clouds: Chamfer 1M-vs-1M fp32
        n = 1_000_000
        def make(seed):
            rng = np.random.default_rng(seed)
            if cfg == "gauss":
                return rng.normal(0.5, 0.05, (n, 3)).astype(np.float32)
            if cfg == "cluster":
                return np.concatenate([rng.random((n * 9 // 10, 3)), rng.normal(0.5, 0.002, (n // 10, 3))]).astype(np.float32)
            a = rng.random((n, 3)).astype(np.float32); a[7] = [60.0, -40.0, 25.0]
            return a
        x, y = make(1000 + 2 * rank), make(1001 + 2 * rank)
        tx, ty = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        step = lambda: pcu.chamfer_distance(tx, ty)
        units, alg = 2 * n, 48 * n
        name = {"gauss": "chamfer_distance, 1M-vs-1M fp32 Gaussian sigma = 0.05", "cluster": "chamfer_distance, 1M-vs-1M fp32, 10 % of each cloud in a tight cluster (sigma 0.002)",
                "outlier": "chamfer_distance, 1M-vs-1M fp32 uniform with one far outlier (bbox x 60)"}[cfg]
        def check():
            t0 = time.perf_counter(); ch0, cxy0, cyx0 = oracle.chamfer_distance(x, y, return_index=True, kind=kind); t_cpu = time.perf_counter() - t0
            cpu["obj"] = cpu_obj(2 * n, t_cpu, refkind, f"one full chamfer_distance {n}-vs-{n} fp32 call on the GPU step's own pair (3x kd-tree build + OpenMP search per direction "
                                 f"+ the numpy tail; {t_cpu:.2f} s)", cores=cores_knn)
            ch, cxy, cyx = pcu.chamfer_distance(tx, ty, return_index=True)
            return {"idx_equal": bool(np.array_equal(cxy.cpu().numpy(), cxy0) and np.array_equal(cyx.cpu().numpy(), cyx0)),
                    "chamfer_rel": abs(float(step()) - float(ch0)) / float(ch0), "tol": 1e-4, "stats": pcu.last_stats()}
    elif cfg == "normals":          # 1M-point sheet, k = 16
        n, k = 1_000_000, 16
        rng = np.random.default_rng(9 + rank)
        xy = rng.random((n, 2)) * 2 - 1
        p = np.ascontiguousarray(np.concatenate([xy, (0.3 * np.sin(3 * xy[:, :1]) * np.cos(2 * xy[:, 1:2]))], 1).astype(np.float32))
        tp = torch.from_numpy(p).to(dev)
        step = lambda: pcu.estimate_point_cloud_normals_knn(tp, k)
        units, alg = n, n * (12 + 12 + 8)
        name = f"estimate_point_cloud_normals_knn k={k}, {n} fp32 points"
        def check():
            sel = np.random.default_rng(0).choice(n, 20000, replace=False)
            idx, nrm = step()
            _, c = oracle.knn(p[sel], p, k, True, kind=kind)
            a = (p[c] - p[sel][:, None, :]).astype(np.float64)
            t0 = time.perf_counter(); _, sv, vt = np.linalg.svd(a, full_matrices=False); t_svd = time.perf_counter() - t0
            good = (sv[:, 1] - sv[:, 2]) / sv[:, 0] > 1e-2
            dot = np.abs(np.einsum("ij,ij->i", nrm.cpu().numpy()[sel].astype(np.float64), vt[:, 2, :]))
            cpu["obj"] = cpu_obj(20000, t_svd, "port", "numpy batched SVD of 20 000 of the points' neighbourhoods (the plane fits only, without the KNN; the reference's "
                                 "Eigen JacobiSVD is not in the checkout)", cores=1, unit="points/s")
            return {"points_checked": int(good.sum()), "max_1_minus_abs_dot": float((1 - dot[good]).max()), "tol": 1e-6}
    elif cfg == "morton":           # element-wise integer kernel, the HBM-bound extreme
        n = 16_000_000
        pts = np.random.default_rng(rank).integers(-(1 << 20), 1 << 20, (n, 3)).astype(np.int32)
        tp = torch.from_numpy(pts).to(dev)
        step = lambda: pcu.morton_encode(tp)
        units, alg = n, n * 20
        name = f"morton_encode, {n} int32 points"
        def check():
            c = step().cpu().numpy().view(np.uint64)
            mk = "ref" if oracle.have_ref_morton() else "port"
            t0 = time.perf_counter(); c0 = oracle.morton_encode(pts[:2_000_000], mk); t_cpu = time.perf_counter() - t0
            cpu["obj"] = cpu_obj(2_000_000, t_cpu, "reference" if mk == "ref" else "port", "the reference's own MortonCode64 over the first 2 000 000 points, one thread", cores=1, unit="points/s")
            return {"codes_equal": bool(np.array_equal(c[:2_000_000], c0)), "checker": mk}
    elif cfg == "voxel":
        n = 1_000_000
        p = cloud(5 + rank, n, np.float32); tp = torch.from_numpy(p).to(dev)
        vs = 1.0 / 128.0
        mb, xb = tuple(np.min(p, 0) - vs / 2), tuple(np.max(p, 0) + vs / 2)
        step = lambda: pcu.downsample_point_cloud_on_voxel_grid(vs, tp, min_bound=mb, max_bound=xb)
        units, alg = n, n * 12 + 2_000_000
        name = f"downsample_point_cloud_on_voxel_grid, {n} fp32 points, voxel 1/128"
        def check():
            v = step().cpu().numpy()
            t0 = time.perf_counter(); v0, _ = oracle.voxel_downsample(p[:200000], None, [vs] * 3, mb); t_cpu = time.perf_counter() - t0
            v1 = pcu.downsample_point_cloud_on_voxel_grid(vs, p[:200000], min_bound=mb, max_bound=xb)
            cpu["obj"] = cpu_obj(200000, t_cpu, "port", "Python restatement of src/sample_point_cloud.cpp:163-235 over the first 200 000 points, one thread", cores=1, unit="points/s")
            return {"voxels": int(len(v)), "means_bit_equal_on_200k": bool(np.array_equal(v1, v0))}
    elif cfg == "sinkhorn":         # dense 4096 x 4096 cost matrix, 50 iterations
        m = n = 4096
        rng = np.random.default_rng(7 + rank)
        a = rng.random((m, 3)).astype(np.float32); b = rng.random((n, 3)).astype(np.float32)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        M = pcu.pairwise_distances(ta, tb)
        wa = torch.full((m,), 1.0 / m, dtype=torch.float32, device=dev); wb = torch.full((n,), 1.0 / n, dtype=torch.float32, device=dev)
        iters = 50
        step = lambda: pcu.sinkhorn(wa, wb, M, eps=1e-2, max_iters=iters, stop_thresh=0.0)
        units, alg = iters * m * n, iters * 2 * m * n * 4 + m * n * 4 * 2
        name = f"sinkhorn, {m} x {n} fp32 cost matrix, {iters} iterations (unit: matrix elements x iterations)"
        def check():
            P = step().cpu().numpy()
            t0 = time.perf_counter(); P0, _ = oracle.sinkhorn(wa.cpu().numpy()[:1024], wb.cpu().numpy()[:1024], M.cpu().numpy()[:1024, :1024] , 1e-2, 5, 0.0); t_cpu = time.perf_counter() - t0
            Pfull, it = oracle.sinkhorn(wa.cpu().numpy(), wb.cpu().numpy(), M.cpu().numpy(), 1e-2, iters, 0.0)
            cpu["obj"] = cpu_obj(5 * 1024 * 1024, t_cpu, "port", "the reference's numpy sinkhorn (_sinkhorn.py:36-130) on a 1024 x 1024 corner of the cost matrix, 5 iterations",
                                 unit="elements*iterations/s")
            return {"plan_max_rel_err": float(np.abs(P - Pfull).max() / np.abs(Pfull).max()), "tol": 5e-4}
    else:
        bunny = np.load(os.path.join(ROOT, "tests", "golden", "bunny_v.npy")).astype(np.float64)
        f = np.load(os.path.join(ROOT, "tests", "golden", "bunny_f.npy"))
        from conftest import mesh_samples
        S = mesh_samples(bunny, f, 1_000_000, seed=5)
        tb, ts_ = torch.from_numpy(bunny).to(dev), torch.from_numpy(S).to(dev)
        step = lambda: pcu.chamfer_distance(tb, ts_, return_index=True)
        units, alg = len(bunny) + len(S), 2 * 24 * (len(bunny) + len(S)) + 8 * (len(bunny) + len(S))
        name = "chamfer_distance(return_index=True), bunny (2,885 vertices) vs 1M mesh samples, fp64"
        def check():
            ch, cxy, cyx = step()
            t0 = time.perf_counter(); ch0, cxy0, cyx0 = oracle.chamfer_distance(bunny, S, return_index=True, kind=kind); t_cpu = time.perf_counter() - t0
            cpu["obj"] = cpu_obj(len(bunny) + len(S), t_cpu, refkind, f"one full chamfer_distance(return_index=True) call on the config's own clouds (the 2 885-row direction searches on one "
                                 f"thread, the 1M-row direction on all cores; {t_cpu:.2f} s)", cores=cores_knn)
            return {"idx_equal": bool(np.array_equal(cxy.cpu().numpy(), cxy0) and np.array_equal(cyx.cpu().numpy(), cyx0)),
                    "chamfer_rel": abs(float(ch) - float(ch0)) / float(ch0), "tol": 1e-6}
    pcu.set_timing(0)
    step()
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=args.coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # dominant-kernel duration: HIP events around the call's main search launch(es), recorded by the library on its launch stream, in extra
    # steps OUTSIDE the timed region (each event is a bubble between kernels); the batch entry point (c4) records none
    k_ms, k_calls = 0.0, 0
    if rank == 0 and not args.no_kernel_events and cfg in ("c1", "c2", "c3", "c5", "gauss", "cluster", "outlier"):
        pcu.set_timing(1)
        for _ in range(4):
            step()
            st = pcu.last_stats()
            if st["n_kernel_search"] > 0:
                k_ms += st["ms_kernel_search"]; k_calls += 1
        pcu.set_timing(0)
    if rank == 0:
        if not args.no_parity:
            parity = check()
        steps = max(args.steps, 1)
        unit = "query-points/s" if cfg in ("c1", "c2", "c3", "c4", "c5", "gauss", "cluster", "outlier") else ("elements*iterations/s" if cfg == "sinkhorn" else "points/s")
        out = {"metric": f"{unit}, {name}", "value": units * world * steps / dt, "unit": unit, "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64" if cfg in ("c1", "c5") else ("i32" if cfg == "morton" else "f32"), "data": "synthetic",
               "config": {"workload": name + ", inputs resident in HBM", "baseline_config": cfg,
                          **({"collectives": "gloo on host tensors: REHEARSAL, ranks share GPUs (PCU_BENCH_SHARE_GPU)"} if args.share else {})},
               "roofline": config_roofline(cfg, alg, dt / steps, k_ms, k_calls),
               "parity": parity}
        if "obj" in cpu and world == 1:
            out["cpu_baseline"] = cpu["obj"]
        if not emit:
            return out
        print(json.dumps(out), flush=True)
    if not emit:
        return None
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
