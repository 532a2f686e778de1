#!/usr/bin/env python
"""bench.py -- Mdisparities/s of the CStereoMatching hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either as above -- bench.py then starts N ranks itself through torch.distributed.run on 127.0.0.1 -- or
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (ConstructPyrm -> MatchOneLayer x PyrmNum, both directions ->
DisparityToCloud) over a batch of --inflight (3) stereo pairs per GPU, inputs already resident in HBM, followed (N > 1)
by the RCCL fan-in gather of the per-pair clouds to rank 0 (in flight while the next step's pairs are matched; all
gathers complete inside the timed region).  Workload at every N: BASELINE.json
configs[1] = C2 (4096x3072, 5 levels, 11x11 NCC, 128 disparities at the lowest level), differently-seeded pairs on
every rank (weak scaling).  `python bench.py --gpus N` started plainly launches its N ranks itself.

metric value = sum over ranks of V_top (masked view-0 top-level pixels inside the margin) per step
             / (max-over-ranks wall time per step) / 1e6.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=["c2", "c2s", "c1", "c3", "c4", "c4s", "c5"],
                    help="c2 (default): BASELINE configs[1], weak scaling; c4: BASELINE configs[3] = the 10-pair C3 rig sharded pair %% N over "
                         "the N GPUs with the RCCL gather of all ten clouds in every step, STRONG scaling (c4s: the same at the shipped scale)")
    ap.add_argument("--inflight", type=int, default=3, help="pairs in flight per GPU (contexts run concurrently by rsm_run_pairs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="rsm_set_option name=value (tuning knobs)")
    ap.add_argument("--stage-events", type=int, default=0, help="1: per-stage events inside the timed region too")
    ap.add_argument("--measure-traffic", type=int, default=1,
                    help="1 (default, N = 1): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one pair in a child process give "
                         "roofline.traffic; 0: quote profiles/pmc_traffic.json while it still describes this kernel source")
    ap.add_argument("--adapter-inflight", type=int, default=0,
                    help="pairs in flight inside the C++ adapter's MatchAll (0: three more than --inflight: a slot's pair is uploading, being filtered or downloading part of the time; measured on C2 with 36 pairs, plain / with the filter inside: 5 slots 302 / 210, 6 303 / 219, 7 301 / 210 Mdisp/s)")
    ap.add_argument("--adapter-pairs", type=int, default=18,
                    help="pairs matched through the compiled C++ adapter (tests/cpp/adapter_bench.cpp) for value_adapter_pcie_inclusive; 0: skip")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: what carries the per-pair clouds to rank 0 -- rccl (default): the library's own rsm_comm_create + rsm_gather_clouds "
                         "(csrc/rsm_comm.hip: what a C++ pipeline links, the replacement of CCloudOptimization.cpp:61,123), posted by a gather thread per rank; "
                         "torch: torch.distributed's batched isend / irecv (reconstruction_amd/dist.py).  The gloo stand-in of the tests always uses torch")
    ap.add_argument("--one-process", action="store_true",
                    help="with --gpus N: ONE process drives the N GPUs through rsm_match_pairs_multi_gpu (contexts on different devices, host buffers in "
                         "and out, no RCCL): an auxiliary PCIe-inclusive figure, not the driver's scaling run")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.one_process:
        raise SystemExit(one_process(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started plainly: become the launcher of N ranks (one process per GPU)
        raise SystemExit(self_launch(args.gpus))

    from reconstruction_amd import Context, run_pairs, synth
    from reconstruction_amd.dist import gather_clouds_async

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- refusing to report a %d-GPU number as --gpus %d"
                         % (args.gpus, world, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # RSM_BENCH_BACKEND=gloo lets the N > 1 control flow be exercised on a single-GPU box (ranks share device 0,
    # clouds go through host memory); the real multi-GPU run uses nccl = RCCL over xGMI.
    backend = os.environ.get("RSM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d needs %d visible GPUs, found %d" % (world, world, torch.cuda.device_count()))

    make = {"c2": synth.config_c2, "c2s": synth.config_c2_sample, "c1": synth.config_c1,
            "c3": synth.config_c3, "c4": synth.config_c3, "c4s": synth.config_c3_shipped, "c5": synth.config_c5}[args.config]
    rig = args.config in ("c4", "c4s")   # BASELINE configs[3]: ONE fixed job -- the rig's ten pairs -- split over the ranks
    N_RIG = 10
    dev = torch.device("cuda", local_rank)
    ctxs, cfgs, keep = [], [], []
    jobs = None
    if rig:
        mine = [p for p in range(N_RIG) if p % world == rank]          # pair % n_gpus -> 2,2,1,1,1,1,1,1 at N = 8
        F = max(1, min(args.inflight, len(mine)))
        jobs = [[] for _ in range(F)]                                   # context i matches its pairs one after the other
        for j, p in enumerate(mine):
            cfgp = make(pair=p)
            t_img = [torch.from_numpy(np.ascontiguousarray(cfgp.image[v])).to(dev) for v in range(2)]
            t_msk = [torch.from_numpy(np.ascontiguousarray(cfgp.mask[v])).to(dev) for v in range(2)]
            jobs[j % F].append((p, cfgp, t_img, t_msk))                 # the inputs of every local pair stay resident in HBM
        torch.cuda.synchronize()
        for i in range(F):
            c = Context(local_rank)
            for o in args.opt:
                k, v = o.split("=")
                c.set_option(k, int(v))
            if jobs[i]:
                p, cfgp, t_img, t_msk = jobs[i][0]
                c.upload_pair_device(cfgp, [t.data_ptr() for t in t_img], [t.data_ptr() for t in t_msk])
                cfgs.append(cfgp)
            ctxs.append(c)
        if not cfgs:                                                    # a rank without a pair (N > 10) only joins the collectives
            cfgs.append(make(pair=0))
    else:
        F = max(1, args.inflight)
        for i in range(F):  # a differently-seeded pair per rank and slot, uploaded once, outside the timed region
            cfg = make(pair=rank * F + i)
            c = Context(local_rank)
            for o in args.opt:
                k, v = o.split("=")
                c.set_option(k, int(v))
            t_img = [torch.from_numpy(np.ascontiguousarray(cfg.image[v])).to(dev) for v in range(2)]
            t_msk = [torch.from_numpy(np.ascontiguousarray(cfg.mask[v])).to(dev) for v in range(2)]
            torch.cuda.synchronize()
            c.upload_pair_device(cfg, [t.data_ptr() for t in t_img], [t.data_ptr() for t in t_msk])
            ctxs.append(c); cfgs.append(cfg); keep.append((t_img, t_msk))
    ctx, cfg = ctxs[0], cfgs[0]
    # N > 1: the transport of the per-pair clouds.  rccl = the product's own path (rsm_comm_create + rsm_gather_clouds on a
    # communicator of its own; the 128-byte id travels through torch.distributed's store, a side channel only)
    # RSM_BENCH_SELF_GATHER=1 (tests): at N = 1 the step still packs its clouds and hands them to the gather thread, which runs
    # rsm_gather_clouds on a one-rank communicator -- the whole N > 1 control flow of this file on a one-GPU box
    exchange = world > 1 or os.environ.get("RSM_BENCH_SELF_GATHER") == "1"
    transport = args.transport if (exchange and backend == "nccl") else "torch"
    comm = None
    transport_note = None
    if exchange and transport == "rccl":
        from reconstruction_amd.dist import Comm
        err = None
        try:
            ids = [Comm.unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001  (librccl not loadable)
            ids, err = [None], e
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None and err is None:
            try:
                comm = Comm(ids[0], rank, world, local_rank)
            except Exception as e:  # noqa: BLE001
                err = e
        else:
            err = err or RuntimeError("rank 0 could not make the communicator id")
        # every rank must take the same transport: one failure anywhere sends all of them to torch.distributed (and the line says so)
        bad = torch.tensor([1.0 if err is not None else 0.0], device=dev)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad[0]) > 0:
            if comm is not None:
                comm.close()
            comm, transport = None, "torch"
            transport_note = "rsm_comm_create failed on some rank (%s): fell back to torch.distributed" % (str(err)[:120] if err else "another rank")
    n_pairs_total = N_RIG if rig else world * F   # pair ids of one step's gather
    if args.pmc_child:  # the profiled child of measure_traffic(): one pair, nothing else
        ctx.run_pair()
        torch.cuda.synchronize()
        for c in ctxs:
            c.close()
        return

    # N > 1: every context free-runs its K steps on its own host thread exactly as at N = 1 (run -> pack the cloud as
    # 16-byte records), handing each step's records to this thread, which posts the fan-in gathers in a fixed order
    # (step, then slot) -- collectives must be posted alike on every rank -- and leaves the gather of step i in flight
    # while step i + 1 is matched; every gather completes inside the timed region.
    import queue
    import threading

    vtop_of = {}   # rig: V_top of every local pair (learnt in the first step)

    def run_steps(k):
        if world == 1 and not rig and not exchange:
            # K steps = every context matches its pair K times.  The contexts are not made to meet between steps: as in
            # rsm_match_pairs (a stream of pairs over a pool of contexts) one pair's launch-bound small levels run under
            # the other's top-level sweeps, also across step boundaries.
            run_pairs(ctxs, repeats=k)
            return
        qs = [queue.Queue() for _ in ctxs]

        def worker(i, c):
            try:
                torch.cuda.set_device(local_rank)
                for _ in range(k):
                    recs = []
                    for p, cfgp, t_img, t_msk in (jobs[i] if rig else [(rank * F + i, None, None, None)]):
                        if rig and len(jobs[i]) > 1:      # several pairs share this context: the next one's images come from HBM (D2D)
                            c.upload_pair_device(cfgp, [t.data_ptr() for t in t_img], [t.data_ptr() for t in t_msk])
                        c.run_pair()
                        if rig and p not in vtop_of:
                            vtop_of[p] = c.download_pair(want_cloud=False, want_disparity=False).v_top
                        if exchange:
                            n = c.n_points
                            rec = torch.empty((n, 16), dtype=torch.uint8, device=dev)  # rsm_point16 records: 16 B per point
                            c.pack_cloud16(rec.data_ptr(), n)
                            recs.append((p, rec if backend == "nccl" else rec.cpu()))
                    qs[i].put(recs)
            except BaseException as e:  # the main thread must not wait for records that never come
                qs[i].put(e)

        threads = [threading.Thread(target=worker, args=(i, c)) for i, c in enumerate(ctxs)]
        for t in threads:
            t.start()
        # rccl transport: rsm_gather_clouds is synchronous, so a gather thread per rank posts the steps' gathers in order
        # (collectives are posted alike on every rank) while the workers match the next step
        gq, gerr = queue.Queue(maxsize=2), []

        def gatherer():
            torch.cuda.set_device(local_rank)
            while True:
                item = gq.get()
                if item is None:
                    return
                try:
                    if not gerr:
                        comm.gather(item, n_pairs_total, root=0)
                except BaseException as e:  # noqa: BLE001
                    gerr.append(e)

        gth = None
        if comm is not None:
            gth = threading.Thread(target=gatherer)
            gth.start()
        pending = None
        for _ in range(k):
            local = []
            for i in range(len(ctxs)):
                recs = qs[i].get()
                if isinstance(recs, BaseException):
                    raise recs
                local += recs
            if comm is not None:
                try:   # at most one gather in flight + one queued: the records stay alive in the queue
                    gq.put(sorted(local, key=lambda t: t[0]), timeout=300)
                except queue.Full:   # a collective that never completes must end the run with a message, not hang it
                    print("bench.py: rsm_gather_clouds made no progress for 300 s on rank %d" % rank, file=sys.stderr, flush=True)
                    os._exit(3)
            elif world > 1:
                h = gather_clouds_async(sorted(local, key=lambda t: t[0]), dst=0)
                if pending is not None:
                    pending.wait()
                pending = h
        if pending is not None:
            pending.wait()
        if gth is not None:
            try:
                gq.put(None, timeout=300)
            except queue.Full:
                print("bench.py: rsm_gather_clouds made no progress for 300 s on rank %d" % rank, file=sys.stderr, flush=True)
                os._exit(3)
            gth.join(timeout=300)
            if gth.is_alive():
                print("bench.py: the last rsm_gather_clouds did not complete within 300 s on rank %d" % rank, file=sys.stderr, flush=True)
                os._exit(3)
            if gerr:
                raise gerr[0]
        for t in threads:
            t.join()

    if args.warmup:
        run_steps(args.warmup)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: only the dominant kernel's launches are bracketed with events (every 8th); the per-stage events
    # cost ~0.35 ms per step, so the stage split comes from one extra, untimed step afterwards
    for c in ctxs:
        c.profile_enable(1 if args.stage_events else 2)
    prof_acc = {}
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    for c in ctxs:
        for k, v in c.profile_get().items():
            a = prof_acc.setdefault(k, {"ms": 0.0, "launches": 0, "bytes": 0.0})
            a["ms"] += v["ms"]; a["launches"] += v["launches"]; a["bytes"] += v["bytes"]
    for c in ctxs:
        c.profile_enable(False)
    ctx.profile_enable(1)
    ctx.run_pair()  # the stage split: one pair alone, untimed
    fence()
    stage_prof = ctx.profile_get()
    ctx.profile_enable(False)

    single = None
    if world == 1 and rank == 0:
        # one context alone, same workload, same run: what ONE pair costs with the GPU to itself (BASELINE configs[1]
        # is a single pair), and the host-buffer boundary (rsm_upload_pair / rsm_download_pair over PCIe) beside it
        ks = max(2, args.steps // 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_pairs([ctx], repeats=ks)
        torch.cuda.synchronize()
        single = (time.perf_counter() - t1) / ks
        t1 = time.perf_counter()
        ctx.upload_pair(cfg)                     # pageable host buffers -> HBM (100 MB at C2)
        h2d = time.perf_counter() - t1
        ctx.run_pair()
        ctx.download_pair()                      # first call sizes and faults the host buffers
        t1 = time.perf_counter()
        ctx.download_pair()                      # both fp64 disparity maps + the cloud (fp64 xyz + BGR)
        d2h = time.perf_counter() - t1
        pres = ctx.download_pair(pinned=True)    # the same into page-locked buffers (rsm_host_alloc), allocated once
        t1 = time.perf_counter()
        ctx.download_pair(into=pres)             # one DMA per buffer
        d2h_pinned = time.perf_counter() - t1
        del pres
        # the per-pair cloud filter on the cloud just made (statistical outlier removal k = 100 / 1 sigma +
        # radius-2.5 normals, CReconstruction.cpp:18), on the GPU, output = the RCCL payload without the outliers
        n_pts = ctx.n_points
        rec = torch.empty((n_pts, 16), dtype=torch.uint8, device=dev)
        nrm = torch.empty((n_pts, 4), dtype=torch.float32, device=dev)
        filt = None
        for _ in range(2):                       # the first call sizes the filter's arena
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_kept, _st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n_pts, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
            torch.cuda.synchronize()
            filt = (time.perf_counter() - t1, n_kept)
        del rec, nrm
    res = ctx.download_pair(want_cloud=False, want_disparity=False)
    if rig:
        v_top = sum(vtop_of.values())        # this rank's pairs of the rig, once per step
    else:
        v_top = sum(c.download_pair(want_cloud=False, want_disparity=False).v_top for c in ctxs)
    if world > 1:
        t = torch.tensor([dt, float(v_top)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        v_total = float(tsum[1])
    else:
        v_total = float(v_top)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = v_total / (dt / args.steps) / 1e6
        # ---- roofline of the dominant kernel: the top level's Jacobi sweep (k_refine_sweep<1>, one launch per sweep)
        # (hipEvents recorded by the library right around every 8th k_refine_sweep<1> launch, on its own stream)
        # which kernel carries the top level's sweeps: k_refine_skew<T,1> (T time-skewed sweeps per launch, the default
        # once the iteration has settled) or the single-sweep k_refine_sweep<1>
        light_b = prof_acc["refine_light_top"]["bytes"] / max(1, prof_acc["refine_light_top"]["launches"])
        dom = "refine_skew_top" if prof_acc.get("refine_skew_top", {"launches": 0})["launches"] > 0 else "refine_light_top"
        top = prof_acc[dom]
        spl = int(round(top["bytes"] / max(1, top["launches"]) / light_b)) if light_b > 0 else 1  # sweeps per launch
        kname = {"refine_skew_top": "k_refine_skew<%d,1> (DisparityRefine, %d time-skewed Jacobi sweeps per launch, top level; template arguments: sweeps per launch, top level)" % (spl, spl),
                 "refine_light_top": "k_refine_sweep<1> (DisparityRefine Jacobi sweep, top level)"}[dom]
        multi = dom != "refine_light_top"
        launches = max(1, top["launches"])
        avg_ms = top["ms"] / launches
        bytes_per_launch = top["bytes"] / launches  # 16 B x 2 directions x P_top
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass, and neither can
        # share a run with the timing) of one pair in a child process, summed per dispatch of the dominant kernel and
        # corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2).  If the profiler is not usable here,
        # the committed summary is quoted instead -- only while it still describes THIS kernel source and workload.
        traffic, traffic_source, traffic_detail = None, None, None
        # (not under a profiler of its own: a rocprofv3 inside a rocprofv3 run is not a supported arrangement)
        profiled = any(k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if world == 1 and args.measure_traffic and profiled:
            traffic_detail = {"skipped": "bench.py itself runs under a profiler"}
        if world == 1 and args.measure_traffic and not profiled:
            try:
                traffic_detail = measure_traffic(kname.split(" ")[0].rstrip(">"), args)
                traffic = traffic_detail["traffic_bytes_per_launch"]
                traffic_source = "measured"
            except Exception as e:  # noqa: BLE001
                traffic_detail = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if traffic is None:
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pj = json.load(f)
                if pj.get("kernel_src_sha256") == kernel_src_sha() and pj.get("workload") == cfg.name and pj.get("kernel") in (kname.split(" ")[0], kname.split(" ")[0].rsplit(",", 1)[0] + ","):
                    traffic = pj["traffic_bytes_per_launch"]
                    traffic_source = "quoted"
            except Exception:
                traffic = None
        stage_ms = {k: round(v["ms"], 3) for k, v in stage_prof.items()}  # one untimed step with stage events
        # the same kernel with the GPU to itself (the untimed single-pair run): what the other pair in flight costs it
        alone = stage_prof[dom]
        alone_ms = alone["ms"] / max(1, alone["launches"])
        alone_gbs = (alone["bytes"] / max(1, alone["launches"])) / (alone_ms * 1e-3) / 1e9 if alone_ms > 0 else 0.0
        total_alg_bytes = sum(v["bytes"] for k, v in stage_prof.items() if k not in ("refine_light_top", "refine_skew_top"))
        out = {
            "metric": "Mdisparities/s per GPU (11x11 NCC, 128 disp)" if args.config == "c2" else "Mdisparities/s",
            "value": round(value, 3), "unit": "Mdisparities/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_pair": round(ms_per_step / F, 3), "higher_is_better": True,
            "scaling": "strong" if rig else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C4_rig10_" + cfg.name.rsplit("_p", 1)[0]) if rig else cfg.name, "width": cfg.width, "height": cfg.height,
                       "pyr_levels": cfg.pyr_levels,
                       "ncc_window": 2 * cfg.radius + 1, "offset": cfg.offset, "pairs_per_gpu": F, "pairs_in_flight": F,
                       "v_top_per_pair": int(res.v_top), "n_points_last": int(res.n_points),
                       "parallelism": ("pairs sharded one process per GPU + fan-in gather of the clouds to rank 0, overlapped with the next step, through %s"
                                       % ("rsm_comm / rsm_gather_clouds (the library's own RCCL path)" if transport == "rccl" else
                                          "torch.distributed (%s)" % backend)) if world > 1 else "single GPU",
                       "transport": transport if exchange else None, "transport_note": transport_note},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         # the same launch priced on its MEASURED HBM bytes (PMC): how close the kernel runs to the
                         # memory system's rate, as opposed to how many of those bytes the algorithm needs
                         "traffic_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic and avg_ms > 0 else None,
                         "traffic_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_ms > 0 else None,
                         "avg_launch_ms": round(avg_ms, 5),
                         "alone": {"avg_launch_ms": round(alone_ms, 5), "achieved": round(alone_gbs, 2),
                                   "frac": round(alone_gbs / HBM_PEAK_GBS, 4),
                                   "note": "one pair in flight (untimed extra run): no other pair's kernels beside the launch"},
                         "launches_timed_per_step": launches // args.steps // F, "concurrent_pairs": F,
                         "sweeps_per_launch": spl,
                         # the first sweeps of the level (cache still filling) run one per launch in k_refine_sweep<1>
                         "single_sweep_kernel": {"kernel": "k_refine_sweep<1>", "avg_launch_ms": round(prof_acc["refine_light_top"]["ms"] / max(1, prof_acc["refine_light_top"]["launches"]), 5),
                                                 "alone_avg_launch_ms": round(stage_prof["refine_light_top"]["ms"] / max(1, stage_prof["refine_light_top"]["launches"]), 5)} if multi else None,
                         "sweeps_per_step": stage_prof["refine_sweep_top"]["launches"],
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "whole_pair_algorithmic_GB": round(total_alg_bytes / 1e9, 3),
                         "whole_pair_frac": round(total_alg_bytes * F / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
            "stage_ms_per_step": stage_ms,
        }
        out["roofline"]["traffic_source"] = traffic_source  # measured = this run's own --pmc passes; quoted = profiles/pmc_traffic.json
        valu = traffic_detail.pop("valu", None) if isinstance(traffic_detail, dict) else None
        scalar = traffic_detail.pop("scalar", None) if isinstance(traffic_detail, dict) else None
        out["roofline"]["traffic_detail"] = traffic_detail
        out["roofline"]["valu"] = valu
        out["roofline"]["scalar"] = scalar
        # What bounds the kernel, from the data: every unit's busy share of the launch, first-class fields beside `frac`
        # (`achieved` / `peak` / `frac` stay the contract's: algorithmic bytes against the 8 TB/s HBM peak = frac_hbm).
        #   frac_hbm_traffic: the MEASURED HBM bytes of a launch against the rate a streaming kernel reaches on this part (6.3 TB/s,
        #                     MI355X_MICROARCH.md), alone as the PMC passes run it
        #   frac_fp64_issue:  the share of the launch a SIMD's VALU is busy (the fp64 issue floor / launch time)
        #   frac_scalar:      scalar instructions per CU and clock (a CU has ONE scalar unit for its four SIMDs)
        #   frac_lds:         the share of the launch a CU's waves are inside LDS instructions
        # `bound` names the largest.  `frac` itself uses roofline.avg_launch_ms (hipEvents around every 8th launch of the timed region,
        # `launches_timed` of them, pairs in flight); profiles/r06_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same
        # command) carries the same kernel's average over all its calls.
        rl = out["roofline"]
        rl["frac_hbm"] = rl["frac"]
        rl["launches_timed"] = launches
        rl["frac_source"] = "algorithmic_bytes_per_launch / avg_launch_ms (hipEvents around %d launches of the timed region, %d pairs in flight) / 8 TB/s" % (launches, F)
        if valu and valu.get("active_frac") is not None and valu.get("launch_ms_in_this_pass"):
            fr = {"fp64_issue": valu["active_frac"]}
            if traffic:
                fr["hbm_traffic"] = round(traffic / (valu["launch_ms_in_this_pass"] * 1e-3) / 1e9 / 6300.0, 4)
            if scalar and scalar.get("scalar_issue_frac") is not None:
                fr["scalar"] = scalar["scalar_issue_frac"]
                fr["lds"] = scalar.get("active_inst_lds_frac_per_cu")
            rl["frac_fp64_issue"] = fr["fp64_issue"]
            rl["frac_hbm_traffic"] = fr.get("hbm_traffic")
            rl["frac_scalar"] = fr.get("scalar")
            rl["frac_lds"] = fr.get("lds")
            best = max((v, k) for k, v in fr.items() if v is not None)
            rl["bound"] = {"fp64_issue": "valu", "hbm_traffic": "hbm", "scalar": "scalar", "lds": "lds"}[best[1]]
            parked = (valu.get("wave_cycles_split") or {}).get("parked_waitcnt_or_barrier")
            rl["bound_note"] = ("busiest unit of the launch (one pair alone, PMC passes): %s at %.0f %%; fp64 VALU %.0f %%, measured HBM traffic %s of the achievable "
                                "6.3 TB/s, scalar unit %s, LDS %s; a wave parked (s_waitcnt / barrier) %s of its cycles -- no unit is saturated: the kernel is bound "
                                "by its waves' dependent chains at five waves per SIMD (LDS and registers allow no more) and by the launch's low-occupancy tail"
                                % (best[1], 100 * best[0], 100 * fr["fp64_issue"],
                                   ("%.0f %%" % (100 * fr["hbm_traffic"])) if fr.get("hbm_traffic") is not None else "?",
                                   ("%.0f %%" % (100 * fr["scalar"])) if fr.get("scalar") is not None else "?",
                                   ("%.0f %%" % (100 * fr["lds"])) if fr.get("lds") is not None else "?",
                                   ("%.0f %%" % (100 * parked)) if parked is not None else "?"))
        if rig:
            # BASELINE configs[3]: the SAME ten pairs at every N (strong scaling): pair p on rank p % N; a step
            # = all ten pairs matched once + (N > 1) the RCCL fan-in of the ten clouds to rank 0; 10 pairs on 8 GPUs cap at 5x
            ppr = [len([p for p in range(N_RIG) if p % world == r]) for r in range(world)]
            out["config"].update({"pairs": N_RIG, "pairs_per_rank": ppr, "pairs_in_flight": [min(args.inflight, max(1, n)) for n in ppr],
                                  "parallelism": "10 pairs, pair %% %d -> rank; per-pair clouds gathered to rank 0 in every step%s"
                                                 % (world, (" through " + ("rsm_gather_clouds (RCCL)" if transport == "rccl" else "torch.distributed")) if world > 1 else " (N = 1: no exchange)"),
                                  "speedup_cap": N_RIG / max(ppr)})
            out["ms_per_pair"] = round(ms_per_step / N_RIG, 3)
        if single is not None:
            out["value_single_pair"] = round(res.v_top / single / 1e6, 3)   # one pair in flight, inputs resident in HBM
            out["ms_single_pair"] = round(single * 1e3, 3)
            out["h2d_ms"] = round(h2d * 1e3, 2)
            out["d2h_ms"] = round(d2h * 1e3, 2)
            out["value_single_pair_pcie_inclusive"] = round(res.v_top / (single + h2d + d2h) / 1e6, 3)
            out["d2h_ms_pinned"] = round(d2h_pinned * 1e3, 2)   # into page-locked buffers the caller keeps (rsm_host_alloc)
            out["value_single_pair_pcie_inclusive_pinned"] = round(res.v_top / (single + h2d + d2h_pinned) / 1e6, 3)
            out["filter_ms"] = round(filt[0] * 1e3, 2)
            out["filter_points"] = {"in": int(n_pts), "kept": int(filt[1])}
        if world == 1 and args.config in ("c2", "c5"):
            # the NCC kernel alone (pixel x candidate evaluations per second) at SURVEY 8(d)'s two points, two launches each
            out["ncc_kernel"] = {}
            for r_, cands in ((5, 129), (7, 257)):
                ms = ctx.bench_ncc(cfg.width, cfg.height, r_, cands, iters=2)
                px = (cfg.width - 2 * r_) * (cfg.height - 2 * r_)
                out["ncc_kernel"]["%dx%d_%d" % (2 * r_ + 1, 2 * r_ + 1, cands)] = {
                    "ms_per_launch": round(ms, 3), "MDE_per_s": round(px * cands / ms / 1e3, 1)}
        if world == 1 and args.config in ("c2", "c3", "c5") and args.adapter_pairs > 0 and not rig:
            # the drop-in as a maintainer integrates it: the compiled C++ adapter's MatchAll (include/rsm_stereo_adapter.hpp),
            # host images in, InsertPoint stream out, PCIe included -- never `value`
            try:
                ab = adapter_bench(cfgs, args.adapter_pairs, args.adapter_inflight if args.adapter_inflight > 0 else F + 3, int(res.v_top))
                out["adapter"] = ab["records16"]
                out["value_adapter_pcie_inclusive"] = ab["records16"]["value"]
                # ... and once the loop's pipeline runs (from the first pair's replay to the last one's: the job's fill left out)
                if "value_steady" in ab["records16"]:
                    out["value_adapter_steady"] = ab["records16"]["value_steady"]
                # the same loop with CCloudOptimization::filter's first half (CCloudOptimization.cpp:82-121, once per pair inside
                # MatchAllLayer, CStereoMatching.cpp:31) on the pair's GPU, inside the loop: MatchAllFiltered
                out["adapter_with_filter"] = ab["gpu_filter"]
                out["value_with_filter"] = ab["gpu_filter"]["value"]
                if "value_steady" in ab["gpu_filter"]:
                    out["value_with_filter_steady"] = ab["gpu_filter"]["value_steady"]
                out["adapter_fp64_points"] = ab["fp64"]
            except Exception as e:  # noqa: BLE001
                out["adapter"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth)
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def one_process(args):
    """`bench.py --gpus N --one-process`: the single-process form of SURVEY 8(e) -- rsm_match_pairs_multi_gpu drives N GPUs
    (context i on GPU i % N, --inflight contexts per GPU) over a queue of C2 pairs with HOST buffers in and out (pageable
    images, the clouds as fp64 xyz + BGR): PCIe-inclusive by construction, no collective.  One JSON line; `value` here is
    NOT the resident-input figure of the default mode and says so."""
    from reconstruction_amd import synth
    from reconstruction_amd.api import match_pairs_multi_gpu
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        print("bench.py: --gpus %d --one-process needs %d visible GPUs, found %d" % (args.gpus, args.gpus, have), file=sys.stderr)
        return 2
    make = {"c2": synth.config_c2, "c2s": synth.config_c2_sample, "c1": synth.config_c1, "c3": synth.config_c3, "c4": synth.config_c3,
            "c4s": synth.config_c3_shipped, "c5": synth.config_c5}[args.config]
    per = max(1, args.inflight)
    distinct = [make(pair=p) for p in range(min(3, args.gpus * per))]
    n_pairs = args.gpus * per * max(1, args.steps)
    cfgs = [distinct[p % len(distinct)] for p in range(n_pairs)]
    match_pairs_multi_gpu(cfgs[:args.gpus * per], args.gpus, per, want_cloud=True, want_disparity=False)   # warm-up (contexts are per call)
    t0 = time.perf_counter()
    res, status, rc = match_pairs_multi_gpu(cfgs, args.gpus, per, want_cloud=True, want_disparity=False)
    dt = time.perf_counter() - t0
    if rc != 0 or any(status):
        print("bench.py --one-process: rc %d, statuses %s" % (rc, status), file=sys.stderr)
        return 1
    v_total = float(sum(r.v_top for r in res))
    print(json.dumps({"metric": "Mdisparities/s", "value": round(v_total / dt / 1e6, 3), "unit": "Mdisparities/s", "n_gpus": args.gpus,
                      "steps": args.steps, "warmup": 1, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": distinct[0].name, "pairs": n_pairs, "pairs_in_flight_per_gpu": per,
                                 "parallelism": "ONE process, rsm_match_pairs_multi_gpu: %d GPUs x %d contexts, host buffers in and out (PCIe-inclusive: "
                                                "pageable images up, fp64 clouds down; contexts created inside the timed call)" % (args.gpus, per)},
                      "note": "auxiliary mode: inputs are NOT resident in HBM; the driver's scaling run is `bench.py --gpus N` (one process per GPU)"}), flush=True)
    return 0


def self_launch(n):
    """Re-executes this command under torch.distributed.run with N ranks on this node (rendezvous on 127.0.0.1) and
    passes the ranks' output through: rank 0 prints the ONE JSON line.  Fewer than N visible GPUs is an error, never a
    silent 1-GPU measurement (RSM_BENCH_BACKEND=gloo, the single-GPU stand-in of the tests, shares device 0)."""
    import socket
    import subprocess
    if os.environ.get("RSM_BENCH_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print("bench.py: --gpus %d needs %d visible GPUs, found %d" % (n, n, have), file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pmc_pass(counters, kernel, args):
    """One `rocprofv3 --pmc <counters> --kernel-trace` pass (nothing else enabled) of `bench.py --pmc-child` (one pair of the
    same workload with the same options).  Returns {counter: (sum over the instances of a dispatch averaged over the
    dispatches of `kernel`, instances per dispatch, dispatches)} and the kernel's average duration in that pass (ms)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    want = kernel.replace(" ", "")
    tmp = tempfile.mkdtemp(prefix="rsm_pmc_", dir="/tmp")
    res, dur = {}, None
    try:
        out = os.path.join(tmp, "p")
        cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--inflight", "1", "--config", args.config, "--no-cpu-baseline"]
        for o in args.opt:
            cmd += ["--opt", o]
        cmd += ["--opt", "refine_split=0"]  # the profiled launches are the timed region's: both directions in one launch
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        per, durs = {}, []
        for path in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
            db = sqlite3.connect(path)
            cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
            name_col = [c for c in cols if c in ("kernel_name", "name")][0]
            for kn, cn, val, did in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % name_col):
                if cn in counters and kn.replace("void ", "").replace(" ", "").startswith(want):
                    e = per.setdefault(cn, {}).setdefault((path, did), [0.0, 0])
                    e[0] += val
                    e[1] += 1
            try:
                for kn, t0, t1 in db.execute("select name, start, end from kernels"):
                    if kn.replace("void ", "").replace(" ", "").startswith(want):
                        durs.append((t1 - t0) / 1e6)
            except Exception:  # noqa: BLE001
                pass
            db.close()
        for cn in counters:
            d = per.get(cn)
            if not d:
                raise RuntimeError("no %s rows for %s" % (cn, kernel))
            res[cn] = (sum(v[0] for v in d.values()) / len(d), sum(v[1] for v in d.values()) / len(d), len(d))
        dur = sum(durs) / len(durs) if durs else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res, dur


def pmc_calibration():
    """FETCH_SIZE / WRITE_SIZE factors measured on known byte counts in the staging pattern of the time-skewed kernel
    (tests/micro/pmc_calib.hip: 8-byte and 2-byte per-lane row loads, 8-byte stores; profiles/pmc_calibration.json);
    MI355X_MICROARCH.md's 2.0 / uncalibrated 1.0 if the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_calibration.json")) as f:
            k = json.load(f)["kernels"]["k_rows"]
        return float(k["read_factor"]), float(k["write_factor"]), "profiles/pmc_calibration.json (k_rows: the kernel's own staging pattern on known bytes)"
    except Exception:  # noqa: BLE001
        return 2.0, 1.0, "MI355X_MICROARCH.md (FETCH_SIZE x 2; WRITE_SIZE uncalibrated)"


def measure_traffic(kernel, args):
    """HBM bytes per launch of `kernel` (a prefix of its name, e.g. k_refine_skew<4,1>): rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, one pass each with --kernel-trace only; counter values summed over the instances of a dispatch,
    averaged over the dispatches; corrected with the calibrated factors of pmc_calibration().  A third pass collects the
    SQ / GRBM counters that say what the kernel is bound by (valu_counters)."""
    rf, wf, src = pmc_calibration()
    f = pmc_pass(["FETCH_SIZE"], kernel, args)[0]["FETCH_SIZE"]
    w = pmc_pass(["WRITE_SIZE"], kernel, args)[0]["WRITE_SIZE"]
    out = {"fetch_size_kib_per_launch": round(f[0], 1), "write_size_kib_per_launch": round(w[0], 1), "dispatches": f[2],
           "correction": "FETCH_SIZE x %.3f, WRITE_SIZE x %.3f: %s" % (rf, wf, src),
           "traffic_bytes_per_launch": (rf * f[0] + wf * w[0]) * 1024.0}
    try:
        out["valu"] = valu_counters(kernel, args)
    except Exception as e:  # noqa: BLE001
        out["valu"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    try:
        out["scalar"] = scalar_counters(kernel, args)
    except Exception as e:  # noqa: BLE001
        out["scalar"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def scalar_counters(kernel, args):
    """The scalar and LDS side of the dominant kernel (one more --pmc pass): a CU has ONE scalar unit for its four SIMDs, so
    scalar wave-instructions are priced per CU, not per SIMD.  SQ_ACTIVE_INST_SCA / SQ_INST_CYCLES_SALU / SQ_ACTIVE_INST_LDS /
    SQ_WAIT_INST_LDS count quad-cycles summed over the chip's waves; the busy share of a unit is that sum x 4 / (units x the
    launch's shader clocks)."""
    ctrs = ["SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
            "GRBM_GUI_ACTIVE"]
    r, dur = pmc_pass(ctrs, kernel, args)
    n_cu = 256
    clocks = r["GRBM_GUI_ACTIVE"][0] / max(1.0, r["GRBM_GUI_ACTIVE"][1])
    if dur and clocks / (dur * 1e6) > 3.0:
        clocks /= 8.0
    sca = r["SQ_ACTIVE_INST_SCA"][0] * 4.0 / n_cu       # clocks in which a CU's waves are inside a scalar instruction
    salu = r["SQ_INST_CYCLES_SALU"][0] * 4.0 / n_cu
    lds = r["SQ_ACTIVE_INST_LDS"][0] * 4.0 / n_cu
    return {"insts_salu": int(r["SQ_INSTS_SALU"][0]), "insts_smem": int(r["SQ_INSTS_SMEM"][0]), "insts_branch": int(r["SQ_INSTS_BRANCH"][0]),
            # lower bound of the scalar unit's busy share: one scalar instruction per clock and CU
            "scalar_issue_frac": round((r["SQ_INSTS_SALU"][0] + r["SQ_INSTS_SMEM"][0]) / n_cu / clocks, 4) if clocks > 0 else None,
            "active_inst_sca_frac_per_cu": round(sca / clocks, 4) if clocks > 0 else None,
            "inst_cycles_salu_frac_per_cu": round(salu / clocks, 4) if clocks > 0 else None,
            "active_inst_lds_frac_per_cu": round(lds / clocks, 4) if clocks > 0 else None,
            "wait_inst_lds_quad_cycles": int(r["SQ_WAIT_INST_LDS"][0]),
            "launch_ms_in_this_pass": round(dur, 5) if dur else None}


def valu_counters(kernel, args):
    """What the dominant kernel's issue side looks like (one more --pmc pass): wave-instructions per launch, the share of
    the launch in which a SIMD's VALU is busy, and the launch time that share alone would take (the fp64 issue floor: the
    kernel cannot run faster than its VALU work back to back).  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_* count
    quad-cycles summed over the chip (MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE counts shader clocks per XCD."""
    ctrs = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "GRBM_GUI_ACTIVE"]
    r, dur = pmc_pass(ctrs, kernel, args)
    n_simd = 256 * 4
    clocks = r["GRBM_GUI_ACTIVE"][0] / max(1.0, r["GRBM_GUI_ACTIVE"][1])      # shader clocks of one launch (mean over the XCDs)
    if dur and clocks / (dur * 1e6) > 3.0:                                       # this rocprofv3 reports ONE row holding the sum over the 8 XCDs
        clocks /= 8.0
    active = r["SQ_ACTIVE_INST_VALU"][0] * 4.0 / n_simd                           # clocks a SIMD's VALU is busy, chip average
    frac = active / clocks if clocks > 0 else None
    wc = max(1.0, r["SQ_WAVE_CYCLES"][0])
    return {"insts_valu": int(r["SQ_INSTS_VALU"][0]), "insts_salu": int(r["SQ_INSTS_SALU"][0]),
            "salu_per_valu": round(r["SQ_INSTS_SALU"][0] / max(1.0, r["SQ_INSTS_VALU"][0]), 3),
            "active_frac": round(frac, 4) if frac is not None else None,
            # the rate dependent fp64 fma chains saturate at on this chip whatever the occupancy or the chains per wave:
            # 0.72-0.75 of 256 CU x 4 SIMD x 16 lanes x 2.4 GHz (tests/micro/ilp_probe.hip, profiles/r05_ilp_probe.log) --
            # the ceiling a VALU-bound fp64 kernel is priced against
            "fp64_issue_ceiling_of_nominal": 0.735,
            "active_frac_of_ceiling": round(frac / 0.735, 4) if frac is not None else None,
            "launch_ms_in_this_pass": round(dur, 5) if dur else None,
            "fp64_issue_floor_ms": round(dur * frac, 5) if dur and frac is not None else None,
            "shader_clock_GHz": round(clocks / (dur * 1e6), 3) if dur else None,
            "wave_cycles_split": {"issuing": round(r["SQ_ACTIVE_INST_ANY"][0] / wc, 3), "parked_waitcnt_or_barrier": round(r["SQ_WAIT_ANY"][0] / wc, 3),
                                  "stalled_for_the_pipe": round(r["SQ_WAIT_INST_ANY"][0] / wc, 3)}}


def adapter_bench(cfgs, n_pairs, inflight, v_top):
    """Builds tests/cpp/adapter_bench.cpp with g++ against the in-tree librsm_mi355.so and lets the C++ adapter's MatchAll
    (RsmStereoAdapter, include/rsm_stereo_adapter.hpp: the pair loop of CStereoMatching.cpp:17-33 with pairs in flight,
    page-locked result buffers, no disparity download) match `n_pairs` pairs -- this run's differently seeded pairs,
    cycled -- from pageable host images to the InsertPoint stream.  Mdisparities/s = n_pairs * V_top / wall time."""
    import shutil
    import subprocess
    import tempfile
    from reconstruction_amd import _lib
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("no g++ on this box")
    tmp = tempfile.mkdtemp(prefix="rsm_adapter_", dir="/tmp")
    try:
        exe = os.path.join(tmp, "adapter_bench")
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.run([gxx, "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_bench.cpp"),
                        "-o", exe, "-L" + libdir, "-lrsm_mi355", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib",
                        "-Wl,--allow-shlib-undefined", "-pthread"], check=True, capture_output=True, timeout=120)
        c0 = cfgs[0]
        with open(os.path.join(tmp, "in.bin"), "wb") as f:
            f.write(np.array([len(cfgs), c0.width, c0.height, c0.pyr_levels, c0.radius, c0.offset, c0.origin_width or c0.width, 0, -1], np.int32).tobytes())
            f.write(np.array([c0.ws], np.float64).tobytes())
            for c in cfgs:
                f.write(np.asarray(c.Q, np.float64).tobytes() + np.asarray(c.R_final, np.float64).tobytes() + np.asarray(c.T_final, np.float64).tobytes())
                for a in (c.image[0], c.image[1], c.mask[0], c.mask[1]):
                    f.write(np.ascontiguousarray(a, np.uint8).tobytes())
        env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        what = {"records16": "RsmStereoAdapter::MatchAll compiled from include/rsm_stereo_adapter.hpp: %d pairs, %d in flight, pageable host images in, "
                             "the cloud as 16-byte records (InsertPoint's float xyz + BGR, packed on the GPU) into page-locked buffers, InsertPoint "
                             "((double)float -> float push_back) per point and filter(pair) replayed in pair order; fp64 disparity maps not "
                             "downloaded (the reference never reads them after CStereoMatching.cpp:29)",
                "fp64": "the same with fp64 xyz + BGR (27 B per point) over PCIe: round 4's form (6 pairs)",
                "gpu_filter": "RsmStereoAdapter::MatchAllFiltered: %d pairs, %d in flight; after each pair's match its GPU runs the first half of "
                              "CCloudOptimization::filter (StatisticalOutlierRemoval k = 100 / 1 sigma, radius-2.5 normals, the turn toward CamCenter: "
                              "CCloudOptimization.cpp:82-121) while the other slots' pairs are matched; the surviving points (16-byte records) and their "
                              "normals come down instead of the raw cloud and are copied out per pair"}
        res = {}
        for key, flags, n in (("records16", 0, n_pairs), ("gpu_filter", 2, n_pairs), ("fp64", 1, min(n_pairs, 6))):
            flags |= int(os.environ.get("RSM_ADAPTER_FLAGS_EXTRA", "0"))   # (A/B of the adapter's options, e.g. 4 = no input staging)
            r = subprocess.run([exe, os.path.join(tmp, "in.bin"), str(n), str(inflight), "0", str(flags), "1"], capture_output=True, text=True, env=env, timeout=300)
            if r.returncode != 0:
                raise RuntimeError("adapter_bench (%s) rc %d: %s" % (key, r.returncode, r.stderr[-200:]))
            d = json.loads(r.stdout.strip().splitlines()[-1])
            d["value"] = round(n * v_top / d["seconds"] / 1e6, 3)
            d["ms_per_pair"] = round(d["seconds"] / n * 1e3, 3)
            if d.get("steady_s_per_pair", 0) > 0:  # from the first pair's replay to the last one's: the loop once its pipeline runs
                d["value_steady"] = round(v_top / d["steady_s_per_pair"] / 1e6, 3)
            d["what"] = what[key] % (n, inflight) if "%d" in what[key] else what[key]
            res[key] = d
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_src_sha():
    """Identity of the dominant kernel's source (the GPU box has no git checkout to ask)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("k_refine_skew.hip", "k_refine.hip", "refine_common.h", "rsm_dev.h"):
        with open(os.path.join(ROOT, "reconstruction_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def cpu_baseline(synth):
    """The CPU oracle (a port of the reference's algorithm, OpenMP row loops) on a BOUNDED sample of the same workload, on
    this box's host cores: C2's geometry at a quarter of the area (same window, levels, offset; ~10 s on 16 cores) -- the
    whole-pair rate per masked pixel is size-independent to a few percent (round 3 measured 0.133 on the full pair, 0.14 on
    this sample) -- and a ~6 s single-thread run of a smaller pair for the scalar figure."""
    from oracle import oracle as orc
    cores = orc.effective_cpus()
    cfg = synth.config_c2_sample(pair=0)
    t0 = time.perf_counter()
    r = orc.match_pair(cfg, want_cloud=True, threads=cores)
    dt = time.perf_counter() - t0
    # (a SAMPLE, not the bench pair itself: C2's geometry scaled to a quarter of the area -- same window, levels, offset, half the
    # candidates at the lowest level.  Round 5 tried a 768-row band of the bench's own pair instead: the rows below an empty parent
    # row search the whole margin at every level whatever the band's height, 82 of its 115 s, so a band is NOT proportional to the
    # pair; the whole pair itself takes 41-43 s on the GPU box's 16 cores = 0.133-0.139 Mdisp/s)
    out = {"value": round(r["v_top"] / dt / 1e6, 5), "unit": "Mdisparities/s", "cores": cores, "kind": "port",
           "sample": "SAMPLE %s: one whole pair of the bench workload's geometry at a quarter of its area (5 levels, 11x11 NCC, offset 2, 64 "
                     "instead of 128 candidates at the lowest level), %d masked pixels, %.1f s (refine %.1f s, NCC match %.1f s); the bench pair "
                     "itself measured once: 0.133-0.139 Mdisp/s (41-43 s)" % (cfg.name, r["v_top"], dt, r["refine_seconds"], r["match_seconds"])}
    # the scalar figure ("1 thread"): the same port on ONE thread, on a sample sized for a few seconds
    small = synth.make_pair(640, 480, 3, radius=5, offset=2, pair=0, mask_kind="rect", mask_l0_width=48, border_l0=6,
                            d0_l0=2.0, amp_l0=1.0, name="C2t_640x480_r5_3levels")
    t0 = time.perf_counter()
    r1 = orc.match_pair(small, want_cloud=True, threads=1)
    dt1 = time.perf_counter() - t0
    out["value_1thread"] = round(r1["v_top"] / dt1 / 1e6, 5)
    out["sample_1thread"] = "%s (11x11 NCC, 3 levels), %d masked pixels, %.1f s on 1 thread" % (small.name, r1["v_top"], dt1)
    return out


if __name__ == "__main__":
    main()
