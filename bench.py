#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native variable-projection hot path.

Metric (BASELINE.json): independent fits/sec (double-exp, m=1024, fp64) at 1/2/4/8 GPUs; % HBM roofline.

A "step" is one pass of the hot path over one batch of synthetic input: a complete batched
Levenberg-Marquardt fit (vp_fit) of B independent double-exponential problems per GPU, from the
initial guesses to convergence, with the data already resident in HBM.  Weak scaling: every rank owns
B problems (problems rank*B .. (rank+1)*B-1 of the global synthetic set); the only collective is one
RCCL all-reduce of 4 doubles per step {sum cost, #ok, #failed, sum evaluations}.

Prints ONE JSON line on rank 0 (see DESIGN.md section 5 for every field):
  value      whole-job fits/s over all ranks (max-over-ranks time, barrier + synchronize on both sides); consecutive
             steps alternate over two handles / HIP streams so that the straggler tail of one launch overlaps the
             next step (config.pipelining); config.single_stream holds the same K steps strictly one at a time
  roofline   the stand-alone Phi/dPhi kernel (vp_basis) against the HBM roofline, HIP-event timed live
  roofline_fit  the fused fit kernel: HBM fraction (tiny by design: y is read once per FIT) and the fp64
                vector-ALU fraction that actually bounds it
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, kind "port") on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The steps alternate over two HIP streams (see --streams) next to RCCL's own stream: with ROCm's default of 4
# hardware queues per process two of those streams can land on the same queue and serialise.  Must be set before
# the HIP runtime initialises (torch is imported inside main()).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536,
                    help="problems per GPU: 65536 = BASELINE configs[3]'s per-GPU shard (524288 over 8 GPUs) and the "
                         "north_star 1-GPU headline size; configs[1] (4096) is measured alongside on rank 0")
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--noise", type=float, default=1e-3)
    ap.add_argument("--streams", type=int, default=2,
                    help="handles / HIP streams the steps alternate over (software pipelining of consecutive batches: "
                         "the straggler tail of step k overlaps the bulk of step k+1); 1 = strictly one batch at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import varpro_amd as vp
    from varpro_amd import distributed as vd
    from varpro_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # (test hooks for a box with ONE GPU: VP_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and VP_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device -- the multi-rank control flow can then be exercised there)
    if os.environ.get("VP_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run even N=1 exercises RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("VP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    B, m = args.batch, args.m
    # ---- synthetic input of BASELINE configs[1] (shard `rank` of the global problem set) ----
    first, count = vd.shard_range(world * B, rank, world)  # contiguous block of the global problem set
    assert count == B
    d = synth.double_exp_batch(B, m=m, first_problem=first, noise=args.noise)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev)
    x = torch.from_numpy(d["x"]).to(dev)
    guess = torch.from_numpy(d["tau_guess"]).to(dev)
    bp = vp.BatchProblem(mdl, Y, x=x)  # device-pointer mode on torch's current stream

    def barrier():
        if use_dist:
            dist.barrier()

    # Software pipelining over consecutive steps: every step is ONE complete batched fit of the B problems (+ the
    # device-side summary + the RCCL all-reduce), but step k runs on handle / stream k mod NS, so that the tail of a
    # launch -- a few fits that need >100 LM iterations while the rest of the GPU is already idle -- overlaps the
    # bulk of the next step's launch.  Each handle owns its own copy of the state; nothing is shared or skipped.
    ns = max(1, args.streams)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(ns - 1)]
    handles = [bp]
    for st in streams[1:]:
        with torch.cuda.stream(st):
            handles.append(vp.BatchProblem(mdl, Y, x=x))
    reds = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(ns)]
    torch.cuda.synchronize()

    def step(k, nslots=ns):
        # everything is enqueued asynchronously: the fit kernel, the 4-double batch summary (device side) and
        # the RCCL-over-xGMI all-reduce of those 32 bytes (the scalar LM cost reduction); no host sync per step
        i = k % nslots
        with torch.cuda.stream(streams[i]):
            handles[i].fit(guess, want_coefficients=False)
            handles[i].summary_device(reds[i])
            if use_dist:
                dist.all_reduce(reds[i], op=dist.ReduceOp.SUM)
        return reds[i]

    def timed(nslots):
        for k in range(args.warmup):
            step(k, nslots)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            last_ = step(k, nslots)
        torch.cuda.synchronize()
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, last_

    dt, last = timed(ns)
    dt_single = timed(1)[0] if ns > 1 else dt  # the same K steps strictly one batch at a time, for reference
    last = last.cpu().numpy()
    total_fits = float(world) * B * args.steps
    value = total_fits / dt
    sum_cost, n_ok, n_bad, n_evals = [float(v) for v in last]
    evals_per_fit = n_evals / (world * B)

    # ---- per-kernel durations with HIP events on the launch stream (rank 0 only) ----
    out = None
    if rank == 0:
        # fit kernel alone (no summary / collective): K launches bracketed by events
        for _ in range(2):
            bp.fit(guess, want_coefficients=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            bp.fit(guess, want_coefficients=False)
        e1.record()
        torch.cuda.synchronize()
        fit_ms = e0.elapsed_time(e1) / args.steps
        # Phi/dPhi kernel: writes n_alpha*m + p*m scalars per problem (constant column not materialised)
        n_alpha, p = 2, 2
        phi = torch.empty((B, n_alpha, m), dtype=torch.float64, device=dev)
        dphi = torch.empty((B, p, m), dtype=torch.float64, device=dev)
        reps = max(args.steps, 20)
        for _ in range(3):
            bp.basis(guess, skip_invariant=True, out_phi=phi, out_dphi=dphi)
        e0.record()
        for _ in range(reps):
            bp.basis(guess, skip_invariant=True, out_phi=phi, out_dphi=dphi)
        e1.record()
        torch.cuda.synchronize()
        basis_ms = e0.elapsed_time(e1) / reps
        T = 8
        bytes_phi = B * (T * (m * n_alpha + m * p) + T * 2) + T * m       # 32784 B/problem + grid
        bytes_fit = B * T * (m + 2 + 3 + 2)                                # SURVEY 8(d) B_fit = 8248 B/fit
        gbs_phi = bytes_phi / (basis_ms * 1e-3) / 1e9
        gbs_fit = bytes_fit / (fit_ms * 1e-3) / 1e9
        # algorithmic fp64 flops of the fused fit: per evaluation
        #   exp: 2m x 28 ; QR sweep of [Phi|y|D] (n=3, 3 extra cols): 2m*(5+4+3)*2 ; per fit additionally
        #   jacobian QR ~ 2m*2*3 per accepted step (~ evaluations)
        flops_eval = 2 * m * 28 + 4 * m * (5 + 4 + 3) + 12 * m
        tflops_fit = B * evals_per_fit * flops_eval / (fit_ms * 1e-3) / 1e12
        # BASELINE configs[1] (B = 4096 on one GPU) alongside: same generator, first 4096 problems
        cfg1 = None
        if B >= 4096 and world == 1:
            bp1 = vp.BatchProblem(mdl, Y[:4096].contiguous(), x=x)
            g1 = guess[:4096].contiguous()
            for _ in range(3):
                bp1.fit(g1, want_coefficients=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n1 = max(args.steps, 10)
            red1 = torch.zeros(4, dtype=torch.float64, device=dev)
            for _ in range(n1):
                bp1.fit(g1, want_coefficients=False)
                bp1.summary_device(red1)
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / n1
            # the same batch pipelined over 4 handles / streams: one 4096-problem launch leaves most of the GPU idle
            # (it is bound by the latency of its slowest fit), consecutive batches fill it
            st4 = [torch.cuda.Stream(device=dev) for _ in range(4)]
            h4, r4 = [], []
            for st in st4:
                with torch.cuda.stream(st):
                    h4.append(vp.BatchProblem(mdl, Y[:4096].contiguous(), x=x))
                    r4.append(torch.zeros(4, dtype=torch.float64, device=dev))
            torch.cuda.synchronize()

            def step4(k):
                with torch.cuda.stream(st4[k % 4]):
                    h4[k % 4].fit(g1, want_coefficients=False)
                    h4[k % 4].summary_device(r4[k % 4])

            for k in range(8):
                step4(k)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            n4 = 4 * n1
            for k in range(n4):
                step4(k)
            torch.cuda.synchronize()
            dt4 = (time.perf_counter() - t4) / n4
            for h_ in h4:
                h_.close()
            cfg1 = {"workload": "BASELINE configs[1]: 4096 fits on 1 GPU (one launch is bound by the latency of its "
                                "slowest fit, >100 LM evaluations)", "fits_per_s": 4096 / dt1, "ms_per_step": dt1 * 1e3,
                    "pipelined_4_streams": {"fits_per_s": 4096 / dt4, "ms_per_step": dt4 * 1e3}}
            bp1.close()
        # HBM traffic of the Phi kernel from the committed rocprofv3 PMC passes (profiles/), per launch
        traffic = traffic_fit = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pj.get("batch") == B and pj.get("m") == m:
                traffic = pj["basis_kernel"]["hbm_bytes_per_launch_corrected"]
                traffic_fit = pj["fit_kernel"]["hbm_bytes_per_launch_corrected"]
        except Exception:
            traffic = traffic_fit = None
        out = {
            "metric": "independent fits/sec (double-exp, m=%d, fp64)" % m,
            "value": value,
            "unit": "fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[3] per-GPU shard (= north_star 1-GPU headline): %d independent "
                            "double-exponential+offset fits per GPU, m=%d, n=3, q=2, fp64, noise %.0e, full LM fit to "
                            "convergence per step" % (B, m, args.noise),
                "batch_per_gpu": B, "m": m, "parallelism": "batch-sharded x%d" % world,
                "pipelining": ("%d handles on %d HIP streams, step k on stream k mod %d: the straggler tail of a launch "
                               "overlaps the bulk of the next step" % (ns, ns, ns)) if ns > 1 else "none (one batch at a time)",
                "single_stream": {"fits_per_s": total_fits / dt_single, "ms_per_step": dt_single / args.steps * 1e3},
                "mean_evaluations_per_fit": evals_per_fit, "fits_successful": n_ok, "fits_failed": n_bad,
                "sum_cost": sum_cost,
            },
            "roofline": {
                "kernel": "basis_kernel (vp_basis: stand-alone Phi/dPhi evaluation)",
                "bound": "hbm", "achieved": gbs_phi, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs_phi / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_launch": bytes_phi, "avg_launch_ms": basis_ms,
            },
            "configs1": cfg1,
            "roofline_fit": {
                "kernel": "fit_kernel (vp_fit: device-resident LM, dominant kernel of the timed step)",
                "bound": "fp64_valu", "achieved": tflops_fit, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops_fit / FP64_VALU_PEAK_TFLOPS,
                "hbm_achieved_GBps": gbs_fit, "hbm_frac": gbs_fit / HBM_PEAK_GBS, "traffic": traffic_fit,
                "bytes_per_launch": bytes_fit, "avg_launch_ms": fit_ms, "fits_per_s_kernel_only": B / (fit_ms * 1e-3),
            },
        }

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores, rank 0, N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = O.max_threads()
        pilot_n = min(B, 16 * cores)
        t1 = time.perf_counter()
        O.fit_batch(mdl, d["x"], d["Y"][:pilot_n], d["tau_guess"][:pilot_n], n_threads=cores)
        pilot = time.perf_counter() - t1
        n_cpu = int(min(max(pilot_n, args.cpu_seconds * pilot_n / max(pilot, 1e-6)), 262144))
        dd = d if n_cpu <= B else synth.double_exp_batch(n_cpu, m=m, noise=args.noise)
        t1 = time.perf_counter()
        _a, _c, rep_cpu, _s = O.fit_batch(mdl, dd["x"], dd["Y"][:n_cpu], dd["tau_guess"][:n_cpu], n_threads=cores)
        wall = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": n_cpu / wall, "unit": "fits/s", "cores": cores, "kind": "port",
            "sample": "first %d problems of the same synthetic workload, %d OpenMP threads, %.1f s wall "
                      "(problem construction + initial evaluation included)" % (n_cpu, cores, wall),
            "mean_evaluations_per_fit": float(rep_cpu["n_evals"].mean()),
            "note": "CPU restatement of the reference algorithm (oracle/varpro_oracle.c), not the Rust reference",
        }
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    for h_ in handles:
        h_.close()
    if use_dist:
        barrier()  # the other ranks wait here while rank 0 takes the per-kernel measurements / CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
