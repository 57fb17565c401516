#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native variable-projection hot path.

Metric (BASELINE.json): independent fits/sec (double-exp, m=1024, fp64) at 1/2/4/8 GPUs; % HBM roofline.

A "step" is one pass of the hot path over one batch of synthetic input: a complete batched
Levenberg-Marquardt fit (vp_fit) of B independent double-exponential problems per GPU, from the
initial guesses to convergence, with the data already resident in HBM.  Weak scaling: every rank owns
B problems (problems rank*B .. (rank+1)*B-1 of the global synthetic set); the only collective is one
RCCL all-reduce of 4 doubles per step {sum cost, #ok, #failed, sum evaluations}.

Prints ONE JSON line on rank 0 (DESIGN.md section 6 explains every field):
  value         whole-job fits/s over all ranks: the K steps with TWO BATCHES IN FLIGHT -- step k runs on handle / HIP stream
                k mod 2 (each step still one complete batched fit of its own B problems + the summary + the all-reduce), so the
                straggler tail of one launch overlaps the bulk of the next (max-over-ranks time, barrier + synchronize on
                both sides; kernel trace of the overlap: profiles/r04_pipelined_overlap.json).  This is the throughput a
                stream of batches gets (varpro_amd.FitPipeline is the product-side form of the same schedule)
  config.one_batch_at_a_time   the same K steps strictly one after another on one stream (rounds 1-3 quoted this as `value`;
                the number that per-kernel durations reproduce: roofline_fit.single_launch_* come from this loop)
  roofline      the stand-alone Phi/dPhi kernel (vp_basis) against the HBM roofline, HIP-event timed live
  roofline_fit  the fused fit kernel against the fp64 vector-ALU peak that bounds it (+ its HBM fraction)
  configs0/1/2/4  BASELINE configs[0], [1], [2], [4] measured on rank 0 at N = 1, each with its own roofline
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, kind "port") on the host's physical cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Two HIP streams (pipelined leg) next to RCCL's own stream: with ROCm's default of 4 hardware queues per process two
# of them can land on the same queue and serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# CPU baseline: one OpenMP thread per PHYSICAL core, pinned (read by libgomp when the oracle library is loaded)
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "spread")

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
FP32_VALU_PEAK_TFLOPS = 157.3


def cpu_topology():
    """(model name, sockets, physical cores, logical cpus) of the host from /proc/cpuinfo"""
    model, phys, logical, cores_per_socket = "unknown", set(), 0, 0
    sockets = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                logical += 1
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
                sockets.add(pid)
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                phys.add((pid, cid))
            elif line.startswith("cpu cores"):
                cores_per_socket = max(cores_per_socket, int(line.split(":", 1)[1]))
    except (OSError, ValueError):
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical or 1
    nsock = max(1, len(sockets))
    # (virtualised hosts often report one "core id" per socket: trust the larger of the two counts)
    ncores = max(len(phys), nsock * cores_per_socket) or logical or 1
    # cgroup CPU bandwidth limit of this container ("max" or "<quota> <period>" in cgroup v2; cfs_quota_us in v1): more
    # runnable threads than quota/period are throttled, so that IS the number of host cores this process can use
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return model, nsock, min(ncores, usable), usable, quota


COMPACT_LINE_LIMIT = 4096  # bytes: the driver's record keeps a bounded tail of stdout (round 5's 21 KB line was not parseable)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=96):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1].rstrip() + "~"


def _rl(r):
    """a roofline block with the prose trimmed: the numbers the contract asks for + the kernel's name"""
    if not isinstance(r, dict):
        return None
    o = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms"))
    if "kernel" in r:
        o["kernel"] = _short(r["kernel"].split(" (")[0], 64)
    if "bound" in o:
        o["bound"] = _short(o["bound"].split(" (")[0], 24)
    return o


def _round_floats(o, sig=6):
    if isinstance(o, float):
        return float("%.*g" % (sig, o)) if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _round_floats(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_floats(v, sig) for v in o]
    return o


def compact(out, extra_file=None):
    """The ONE line the driver parses: every field of the bench contract + `roofline` + `cpu_baseline`, numbers only, bounded
    (< COMPACT_LINE_LIMIT bytes whatever the side legs grow to; tests/test_bench_line.py).  Everything else -- every side
    leg with its prose -- goes to the side file named by `extra_file`."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data", "schema") if k in out}
    c["value_definition"] = _short(out.get("value_definition"), 56)
    cfg = out.get("config", {})
    c["config"] = _pick(cfg, ("batch_per_gpu", "m", "batches_in_flight", "world_size", "parallelism", "mean_evaluations_per_fit",
                              "fits_successful", "fits_failed", "sum_cost", "collective_backend", "per_rank_ms_per_step"))
    c["config"]["workload"] = _short(cfg.get("workload"), 100)
    if "one_batch_at_a_time" in cfg:
        c["config"]["one_batch_at_a_time"] = _pick(cfg["one_batch_at_a_time"], ("ms_per_step", "fits_per_s", "per_rank_ms_per_step"))
    r = out.get("roofline")
    if r:
        c["roofline"] = dict(_rl(r), dominant_kernel=r.get("dominant_kernel"))
    rf = out.get("roofline_fit")
    if rf:
        c["roofline_fit"] = dict(_rl(rf), **_pick(rf, ("dominant_kernel", "single_launch_frac", "single_launch_avg_ms", "hbm_frac",
                                                       "valu_issue_frac", "flops_per_evaluation")))
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "single_thread_fits_per_s", "parallel_efficiency",
                                       "cores_x_single_thread_fits_per_s", "extrapolated_single_socket_fits_per_s"))
        c["cpu_baseline"]["sample"] = _short(cb.get("sample"), 96)
    for k in ("gpu_over_cpu", "gpu_over_cores_x_single_thread", "gpu_over_cpu_single_socket_extrapolated"):
        if k in out:
            c[k] = out[k]
    pc = out.get("parity_census")
    if pc:
        c["parity_census"] = _pick(pc, ("problems", "same_success_class", "same_termination_code", "share_evals_within_3",
                                        "share_evals_equal", "sum_evals_device", "sum_evals_oracle", "share_objective_within_1e-6"))
    # the side legs: one number + one roofline fraction each (full blocks in the side file)
    side = {}

    def leg(name, d, keys, roof="roofline"):
        if isinstance(d, dict):
            e = _pick(d, keys)
            rr = d.get(roof)
            if isinstance(rr, dict) and "frac" in rr:
                e["frac"], e["bound"] = rr["frac"], _short(str(rr.get("bound", "")).split(" (")[0].replace("fp64_valu", "f64v"), 7)
                if rr.get("traffic") is not None and rr.get("bytes_per_launch"):
                    e["traffic_x"] = rr["traffic"] / rr["bytes_per_launch"]  # HBM traffic (PMC) over the algorithmic bytes
            if e:
                side[name] = e

    leg("configs0", out.get("configs0"), ("us_per_fit", "evaluations"))
    leg("configs1", out.get("configs1"), ("fits_per_s", "ms_per_step"))
    leg("configs2", out.get("configs2"), ("global_fit_ms", "global_fit_event_ms", "evaluations", "trait_evaluation_ms"), "roofline_fit")
    if isinstance(out.get("configs2"), dict) and isinstance(out["configs2"].get("two_fits_in_flight"), dict):
        side["configs2"]["ms_per_fit_2_in_flight"] = out["configs2"]["two_fits_in_flight"].get("ms_per_fit")
        side["configs2"]["ms_per_fit_4_in_flight"] = (out["configs2"].get("four_fits_in_flight") or {}).get("ms_per_fit")
    leg("configs4", out.get("configs4"), ("fits_per_s", "ms_per_step", "fraction_failed"))
    c3 = out.get("configs3_emulated")
    if isinstance(c3, dict):
        side["configs3_emulated"] = _pick(c3, ("shards", "predicted_efficiency"))
        if isinstance(c3.get("two_batches_in_flight"), dict):
            side["configs3_emulated"]["predicted_efficiency_2_in_flight"] = c3["two_batches_in_flight"].get("predicted_efficiency")
    leg("evaluate_boundary", out.get("evaluate_boundary"), ("ms",))
    leg("external_model", out.get("external_model"), ("ms_phi_dphi_in_r_J_out",))
    for k, v in (out.get("external_model", {}).get("long_problems", {}) or {}).items():
        leg("external_" + k, v, ("ms",))
    leg("external_fit", out.get("external_fit"), ("fits_per_s", "fits_per_s_steps_only", "steps"))
    for k, v in (out.get("streamed_rows") or {}).items():
        leg("streamed_" + k, v, ("fits_per_s", "ms_per_step"))
        if isinstance(v, dict) and "as_caller_evaluated_model" in v:
            leg("streamed_" + k + "_extfit", v["as_caller_evaluated_model"], ())
    leg("generic_fallback", out.get("generic_fallback"), ("fits_per_s", "ms_per_step"))
    if side:
        c["side"] = side
    if "build" in out:
        c["build"] = _pick(out["build"], ("library_bytes", "kernels", "kernels_above_64_spilled_vgprs", "full_build_cpu_minutes", "full_build_wall_s"))
    if extra_file:
        c["extra_file"] = extra_file
    precise = {k: c[k] for k in ("value", "ms_per_step") if k in c}
    precise_cfg = {k: c["config"][k] for k in ("sum_cost", "mean_evaluations_per_fit") if k in c["config"]}
    c = _round_floats(c)
    c.update(_round_floats(precise, 13))  # (the contract's own numbers and the all-reduced totals keep their digits)
    c["config"].update(_round_floats(precise_cfg, 13))
    # hard bound, with a margin for longer host strings on another box: shed the least important parts until the line fits
    # (never the contract fields, `roofline`, `cpu_baseline`)
    limit = COMPACT_LINE_LIMIT - 256

    def fits():
        return len(json.dumps(c, separators=(",", ":"))) < limit
    if not fits():
        c.pop("value_definition", None)
    side_ = c.get("side", {})
    while not fits() and side_:
        side_.pop(next(reversed(side_)))  # side legs from the end
        c["dropped_for_length"] = c.get("dropped_for_length", 0) + 1
    for k in ("side", "build", "parity_census", "gpu_over_cpu_single_socket_extrapolated", "roofline_fit"):
        if fits():
            break
        c.pop(k, None)
        c["dropped_for_length"] = c.get("dropped_for_length", 0) + 1
    return c


def emit(out):
    """write the full result object to bench_full.json (repo root and gpurun_out/, the directory gpurun merges back) and
    return the compact line for stdout"""
    extra = None
    for dn in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            os.makedirs(dn, exist_ok=True)
            with open(os.path.join(dn, "bench_full.json"), "w") as f:
                json.dump(out, f)
            extra = extra or os.path.relpath(os.path.join(dn, "bench_full.json"), ROOT)
        except OSError:
            pass
    line = json.dumps(compact(out, extra), separators=(",", ":"))
    assert len(line) < COMPACT_LINE_LIMIT and "\n" not in line
    return line


# taken NOW: importing torch initialises its OpenMP runtime, which under OMP_PROC_BIND pins the main thread to one
# place -- the affinity mask read afterwards would show a single core
CPU_TOPOLOGY = cpu_topology()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # 100 x ~2.6 ms: a timed region a GPU-busy sampler can see
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536,
                    help="problems per GPU: 65536 = BASELINE configs[3]'s per-GPU shard (524288 over 8 GPUs) and the "
                         "north_star 1-GPU headline size; configs[0,1,2,4] are measured alongside on rank 0 at N = 1")
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--noise", type=float, default=1e-3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the configs[0,1,2,4] side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of ONE run of the CPU baseline sample (two runs)")
    ap.add_argument("--emulate-shards", type=int, default=8,
                    help="N = 1 only: run the G per-GPU shards of BASELINE configs[3] (G x batch problems) one after another "
                         "on this device and report per-shard time / evaluations and the predicted weak-scaling efficiency "
                         "mean/max (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import varpro_amd as vp
    from varpro_amd import distributed as vd
    from varpro_amd import synth
    from varpro_amd import _lib

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
    backend = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("VP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == world

    B, m = args.batch, args.m
    # ---- synthetic input of BASELINE configs[1]/[3] (shard `rank` of the global problem set) ----
    first, count = vd.shard_range(world * B, rank, world)  # contiguous block of the global problem set
    assert count == B
    d = synth.double_exp_batch(B, m=m, first_problem=first, noise=args.noise)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    Y = torch.from_numpy(d["Y"]).to(dev)
    x = torch.from_numpy(d["x"]).to(dev)
    guess = torch.from_numpy(d["tau_guess"]).to(dev)

    def barrier():
        if use_dist:
            dist.barrier()

    def allreduce(t):
        # gloo (single-GPU test hook) reduces on the host; RCCL reduces the 32 bytes on the device over xGMI
        if backend == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    # handle 0 on the current stream; handle 1 on a second stream for the pipelined leg.  Every step is ONE complete
    # batched fit of the B problems (+ the device-side summary + the all-reduce); each handle owns its full state.
    # The second in-flight handle fits a DIFFERENT batch (the next B problems of the global synthetic set after every rank's
    # first block: a pipeline streams different data through its slots).
    streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)]
    handles = [vp.BatchProblem(mdl, Y, x=x)]
    d2 = synth.double_exp_batch(B, m=m, first_problem=world * B + first, noise=args.noise)
    Y2 = torch.from_numpy(d2["Y"]).to(dev)
    guess2 = torch.from_numpy(d2["tau_guess"]).to(dev)
    guesses = [guess, guess2]
    with torch.cuda.stream(streams[1]):
        handles.append(vp.BatchProblem(mdl, Y2, x=x))
    del d2
    reds = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    bp = handles[0]

    fit_events = []  # (start, end) HIP events around the fit of every TIMED step (recorded on the launch stream, no host sync)

    def step(k, nslots, events=None):
        # everything is enqueued asynchronously: the fit kernel, the 4-double batch summary (device side) and
        # the RCCL-over-xGMI all-reduce of those 32 bytes (the scalar LM cost reduction); no host sync per step
        i = k % nslots
        with torch.cuda.stream(streams[i]):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            handles[i].fit(guesses[i], want_coefficients=False)
            if events is not None:
                e1.record()
                events.append((e0, e1))
            handles[i].summary_device(reds[i])
            if use_dist:
                allreduce(reds[i])
        return reds[i]

    per_rank_s = {}

    def timed(nslots):
        for k in range(args.warmup):
            step(k, nslots)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            last_ = step(k, nslots, fit_events if nslots == 1 else None)
        torch.cuda.synchronize()
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            # every rank's own time for the K steps (rank 0 prints them: the weak-scaling loss of this design is the spread of
            # the ranks' step times)
            mine = torch.tensor([dt_], dtype=torch.float64, device="cpu" if backend == "gloo" else dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank_s[nslots] = [float(t_.item()) for t_ in every]
            tmax = torch.tensor([dt_], dtype=torch.float64, device=dev)
            if backend == "gloo":
                h = tmax.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX)
                tmax = h
            else:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, last_

    dt_one, last = timed(1)        # K steps strictly one batch at a time (single-launch durations: fit_events)
    dt, _last2 = timed(2)          # THE timed region: the same K steps, two batches in flight (handle / stream k mod 2)
    last = last.cpu().numpy()
    total_fits = float(world) * B * args.steps
    value = total_fits / dt
    sum_cost, n_ok, n_bad, n_evals = [float(v) for v in last]
    evals_per_fit = n_evals / (world * B)

    out = None
    cfg4_check = None
    if rank == 0:
        T = 8

        def event_ms(fn, reps, warm=2):
            for _ in range(warm):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        def event_ms_each(fn, reps, warm=3):
            """mean duration of INDIVIDUAL launches (an event pair around each one): what a kernel trace reports per
            dispatch.  Back-to-back launches of a kernel made of very short-lived waves start filling the CUs the
            draining launch frees, so elapsed / reps understates the per-dispatch duration (0.295 vs 0.352 ms here)."""
            for _ in range(warm):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a_, b_ in ev:
                a_.record()
                fn()
                b_.record()
            torch.cuda.synchronize()
            return sum(a_.elapsed_time(b_) for a_, b_ in ev) / reps

        def in_flight_ms(make_handle, fit_fn, depth, rounds, warm=1, threads=False):
            """ms per batch with `depth` batches in flight: `depth` handles, each on its own HIP stream, fitted round-robin
            (wall clock, device synchronised on both sides).  threads=True: one host thread per handle, for entry points
            that wait on the host inside the call (the global fit reads its active count back)."""
            strs = [torch.cuda.Stream(device=dev) for _ in range(depth)]
            hs = []
            for st_ in strs:
                with torch.cuda.stream(st_):
                    hs.append(make_handle())
            torch.cuda.synchronize()

            def run_one(i, n):
                with torch.cuda.stream(strs[i]):
                    for _ in range(n):
                        fit_fn(hs[i])

            def run(n):
                if threads:
                    import threading
                    th = [threading.Thread(target=run_one, args=(i, n)) for i in range(depth)]
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                else:
                    for _ in range(n):
                        for i in range(depth):
                            with torch.cuda.stream(strs[i]):
                                fit_fn(hs[i])
            run(warm)
            torch.cuda.synchronize()
            # host threads: the interpreter hands the GIL over every 5 ms by default -- longer than a fit; the fits release it
            # while they wait, but a thread that wakes up must get it back.  Short switch interval, best of three passes (the
            # spread between passes is the host's scheduling, not the device's: 0.57-0.71 ms per fit box to box before)
            passes = 3 if threads else 1
            old_si = sys.getswitchinterval()
            if threads:
                sys.setswitchinterval(1e-4)
            best = None
            for _p in range(passes):
                t0_ = time.perf_counter()
                run(rounds)
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t0_
                best = dt_ if best is None or dt_ < best else best
            sys.setswitchinterval(old_si)
            for h_ in hs:
                h_.close()
            return best * 1e3 / (rounds * depth)

        TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")

        def committed_traffic(key, field="hbm_bytes_per_launch_corrected"):
            """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r0x_pmc_traffic.json; separate
            FETCH_SIZE / WRITE_SIZE passes, FETCH x2 on gfx950) -- NOT measured in this run: see traffic_source"""
            for fn in TRAFFIC_FILES:
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    return pj[key][field]
                except Exception:
                    continue
            return None

        def committed_traffic_grid(kernel, grid):
            """the same for ONE launch shape of a kernel (per_kernel_and_grid entry)"""
            for fn in TRAFFIC_FILES:
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    return pj["per_kernel_and_grid"]["%s@grid%d" % (kernel, grid)]["hbm_bytes_per_launch_corrected"]
                except Exception:
                    continue
            return None

        def traffic_source(key):
            for fn in TRAFFIC_FILES:
                try:
                    if key in json.load(open(os.path.join(ROOT, "profiles", fn))):
                        return "committed PMC: profiles/" + fn
                except Exception:
                    continue
            return None

        def committed_valu_issue(key):
            """VALU issue fraction over the launch from the committed SQ PMC pass (profiles/r0x_fit_kernels_valu_pmc.json)"""
            for fn in ("r06_fit_kernels_valu_pmc.json", "r05_fit_kernels_valu_pmc.json", "r04_fit_kernels_valu_pmc.json", "r03_fit_kernels_valu_pmc.json", "r02_fit_kernels_valu_pmc.json"):
                try:
                    e = json.load(open(os.path.join(ROOT, "profiles", fn)))[key]
                    # wave-level VALU instructions x 4 issue cycles / (1024 SIMDs x launch duration x 2.4 GHz)
                    frac = e["SQ_INSTS_VALU"] * 4.0 / (1024.0 * e["avg_duration_ns_under_pmc"] * 1e-9 * 2.4e9)
                    return frac, "committed PMC: profiles/" + fn
                except Exception:
                    continue
            return None, None

        # ---- per-kernel durations with HIP events on the launch stream ----
        # the fit kernel: the event pairs recorded around vp_fit INSIDE the timed region above (same loop as ms_per_step, so
        # avg_launch_ms <= ms_per_step by construction; the difference is the summary kernel + the all-reduce)
        fit_ms = sum(a_.elapsed_time(b_) for a_, b_ in fit_events) / max(1, len(fit_events))
        n_alpha, p = 2, 2
        phi = torch.empty((B, n_alpha, m), dtype=torch.float64, device=dev)
        dphi = torch.empty((B, p, m), dtype=torch.float64, device=dev)
        basis_ms = event_ms_each(lambda: bp.basis(guess, skip_invariant=True, out_phi=phi, out_dphi=dphi), max(args.steps, 20), 3)
        basis_ms_back_to_back = event_ms(lambda: bp.basis(guess, skip_invariant=True, out_phi=phi, out_dphi=dphi), max(args.steps, 20), 3)
        del phi, dphi
        bytes_phi = B * (T * (m * n_alpha + m * p) + T * 2) + T * m       # 32784 B/problem + grid
        bytes_fit = B * T * (m + 2 + 3 + 2)                                # SURVEY 8(d) B_fit = 8248 B/fit
        gbs_phi = bytes_phi / (basis_ms * 1e-3) / 1e9
        gbs_fit = bytes_fit / (fit_ms * 1e-3) / 1e9
        # algorithmic fp64 flops of one evaluation, SURVEY 8(d): m n_alpha E_exp + 2 m n^2 + 4 m n S + (4n+2) m S q with
        # n = 3, n_alpha = 2, q = 2, S = 1, E_exp = 28 -> 116736 at m = 1024
        flops_eval = m * 2 * 28 + 2 * m * 9 + 4 * m * 3 + (4 * 3 + 2) * m * 2
        tflops_fit = B * evals_per_fit * flops_eval / (fit_ms * 1e-3) / 1e12
        tflops_dev = B * evals_per_fit * flops_eval * args.steps / dt / 1e12   # device-level: the whole timed region
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
            "schema": 3,  # 1 (rounds 1-3): value = one batch at a time; 2 (round 4): two batches in flight on the SAME data;
                          # 3: two batches in flight on two different batches.  config.one_batch_at_a_time keeps definition 1
            "value_definition": "K complete batched fits, two batches in flight (handle / HIP stream k mod 2, two different batches)",
            "config": {
                "workload": "B=%d fits/GPU, m=%d, fp64, double-exp+offset (n=3, q=2), one full LM fit per step" % (B, m),
                "workload_detail": "BASELINE configs[3] per-GPU shard = north_star 1-GPU headline; noise %.0e; two batches in "
                                   "flight (step k on handle / HIP stream k mod 2); every step starts from the initial guesses" % args.noise,
                "batch_per_gpu": B, "m": m, "parallelism": "batch-sharded x%d" % world,
                "batches_in_flight": 2,
                "world_size": dist.get_world_size() if use_dist else 1,
                "collective_backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if use_dist else None,
                "one_batch_at_a_time": {
                    "fits_per_s": total_fits / dt_one, "ms_per_step": dt_one / args.steps * 1e3,
                    "per_rank_ms_per_step": [t_ / args.steps * 1e3 for t_ in per_rank_s.get(1, [dt_one])],
                    "note": "the same K steps strictly one after another on one HIP stream (what rounds 1-3 quoted as `value`): a "
                            "launch ends with its longest fits (> 100 LM evaluations against a mean of 9) running alone, ~0.5 ms "
                            "in which most of the device idles; with two batches in flight that tail overlaps the bulk of the "
                            "next batch (kernel trace: profiles/r04_pipelined_overlap.json)"},
                "per_rank_ms_per_step": [t_ / args.steps * 1e3 for t_ in per_rank_s.get(2, [dt])],
                "mean_evaluations_per_fit": evals_per_fit, "fits_successful": n_ok, "fits_failed": n_bad,
                "sum_cost": sum_cost,
            },
            "roofline": {
                "kernel": "basis_flat_kernel (vp_basis: stand-alone Phi/dPhi evaluation; one thread per row pair of one column: "
                          "Phi and dPhi are each written as one linear sweep)",
                "dominant_kernel": False, "dominant_kernel_roofline": "roofline_fit (fit2_kernel: the timed step is one launch of it; "
                "bound by the fp64 vector ALU, which the roofline contract's hbm / mfma bounds cannot express)",
                "bound": "hbm", "achieved": gbs_phi, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs_phi / HBM_PEAK_GBS, "traffic": committed_traffic("basis_flat_kernel") or committed_traffic("basis_rowpair_kernel"),
                "traffic_source": traffic_source("basis_flat_kernel") or traffic_source("basis_rowpair_kernel"),
                "bytes_per_launch": bytes_phi, "avg_launch_ms": basis_ms, "back_to_back_ms_per_launch": basis_ms_back_to_back,
            },
            "roofline_fit": {
                "kernel": "fit2_kernel (vp_fit: persistent slot kernel, dominant kernel of the timed step)",
                "dominant_kernel": True,
                "bound": "fp64_valu", "achieved": tflops_dev, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops_dev / FP64_VALU_PEAK_TFLOPS, "flops_per_evaluation": flops_eval,
                "what": "achieved = algorithmic fp64 flops of the K timed steps / the timed region (two launches share the device, "
                        "so the device-level rate is the one that can be set against the device's peak); single_launch_* = one "
                        "launch alone on the device, HIP-event pairs around vp_fit inside the one-batch-at-a-time loop (what a "
                        "kernel trace of that loop reports per dispatch)",
                "single_launch_avg_ms": fit_ms, "single_launch_achieved": tflops_fit,
                "single_launch_frac": tflops_fit / FP64_VALU_PEAK_TFLOPS,
                "hbm_achieved_GBps": gbs_fit, "hbm_frac": gbs_fit / HBM_PEAK_GBS, "traffic": committed_traffic("fit2_kernel"),
                "traffic_source": traffic_source("fit2_kernel"),
                "valu_issue_frac": committed_valu_issue("fit2_kernel")[0], "valu_issue_source": committed_valu_issue("fit2_kernel")[1],
                "bytes_per_launch": bytes_fit, "avg_launch_ms": fit_ms, "fits_per_s_kernel_only": B / (fit_ms * 1e-3),
            },
        }

        # ================= side measurements: BASELINE configs[0], [1], [2], [4] (rank 0, N = 1 only) =================
        if world == 1 and not args.no_side_configs:
            # ---- configs[0]: the reference's own bench problem, ONE fit (latency, not throughput) ----
            c0 = synth.config0()
            mdl0 = vp.multi_exponential_model(c0["x"], c0["tau_guess"])
            bp0 = vp.BatchProblem(mdl0, torch.from_numpy(c0["y"][None, :]).to(dev), x=torch.from_numpy(c0["x"]).to(dev))
            g0 = torch.from_numpy(c0["tau_guess"][None, :]).to(dev)
            a0, _c0, rep0 = bp0.fit(g0)
            r0 = bp0.report_to_numpy(rep0)
            us0 = event_ms(lambda: bp0.fit(g0, want_coefficients=False), 50, 5) * 1e3
            out["configs0"] = {
                "workload": "BASELINE configs[0]: benches/double_exponential_without_noise.rs, 1 fit, m=1024 (quirk grid), "
                            "fit only (setup excluded)",
                "us_per_fit": us0, "evaluations": int(r0["n_evals"][0]), "termination": int(r0["termination"][0]),
                "us_per_evaluation": us0 / max(1, int(r0["n_evals"][0])),
                "max_abs_tau_error": float(np.abs(a0.cpu().numpy()[0] - c0["tau_true"]).max()),
                "roofline": {"kernel": "fit_kernel (one wavefront)", "bound": "latency",
                             "note": "one wavefront on one SIMD of 1024: a dependent chain of ~1800 VALU instructions "
                                     "per LM evaluation; no throughput roofline applies to a single fit"},
            }
            bp0.close()

            # ---- configs[1]: B = 4096 on one GPU, one launch at a time ----
            if B >= 4096:
                bp1 = vp.BatchProblem(mdl, Y[:4096].contiguous(), x=x)
                g1 = guess[:4096].contiguous()
                ms1 = event_ms(lambda: bp1.fit(g1, want_coefficients=False), max(args.steps, 20), 3)
                red1 = torch.zeros(4, dtype=torch.float64, device=dev)
                bp1.summary_device(red1)
                ev1 = float(red1.cpu()[3])
                Y1 = Y[:4096].contiguous()
                ms1_q = {dq: in_flight_ms(lambda: vp.BatchProblem(mdl, Y1, x=x), lambda h_: h_.fit(g1, want_coefficients=False),
                                          dq, max(args.steps // 4, 10), 2) for dq in (2, 4, 8)}
                out["configs1"] = {
                    "workload": "BASELINE configs[1]: 4096 fits on 1 GPU, one launch at a time (bound by the latency of "
                                "its slowest fit, >100 LM evaluations; kernel selection automatic = one wavefront per problem)",
                    "fits_per_s": 4096 / (ms1 * 1e-3), "ms_per_step": ms1,
                    "batches_in_flight": {str(dq): {"ms_per_batch": v_, "fits_per_s": 4096 / (v_ * 1e-3),
                                                    "frac_fp64_valu": ev1 * flops_eval / (v_ * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}
                                          for dq, v_ in ms1_q.items()},
                    "roofline": {"kernel": "fit_kernel", "bound": "fp64_valu",
                                 "achieved": ev1 * flops_eval / (ms1 * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": ev1 * flops_eval / (ms1 * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS},
                }
                bp1.close()

            # ---- configs[3] emulated on ONE device: the G per-GPU shards of the G x B problem set, one after another ----
            # The multi-GPU run is weak-scaled with no data-path collective: rank g fits shard g (problems g*B ..
            # (g+1)*B-1) and the step ends with a 32-byte all-reduce, so its step time is the SLOWEST shard's.  The only
            # scaling loss this design has is the imbalance of the shards' work (evaluation counts are heavy-tailed):
            # predicted efficiency = mean / max of the per-shard step times.
            G = args.emulate_shards
            if G > 1:
                shard_ms, shard_evals, shard_max, shard_ms2 = [], [], [], []
                for g in range(G):
                    fg, cg = vd.shard_range(G * B, g, G)
                    dg = d if g == 0 else synth.double_exp_batch(cg, m=m, first_problem=fg, noise=args.noise)
                    if g > 0:
                        bp.set_observations(torch.from_numpy(dg["Y"]).to(dev))
                    gg = torch.from_numpy(dg["tau_guess"]).to(dev)
                    shard_ms.append(event_ms(lambda: bp.fit(gg, want_coefficients=False), 5, 1))
                    _ag, _cg, repg = bp.fit(gg, want_coefficients=False)
                    ng = bp.report_to_numpy(repg)["n_evals"]
                    shard_evals.append(int(ng.sum()))
                    shard_max.append(int(ng.max()))
                    # the same shard the way a rank of the real run steps: TWO batches in flight (handle / stream k mod 2; the
                    # second handle fits the rank's block of the NEXT G x B problems, as in the timed region above)
                    dg2 = synth.double_exp_batch(cg, m=m, first_problem=G * B + fg, noise=args.noise)
                    with torch.cuda.stream(streams[1]):
                        handles[1].set_observations(torch.from_numpy(dg2["Y"]).to(dev))
                    gg2 = torch.from_numpy(dg2["tau_guess"]).to(dev)
                    torch.cuda.synchronize()

                    def two_steps(nsteps):
                        for k_ in range(nsteps):
                            with torch.cuda.stream(streams[k_ % 2]):
                                handles[k_ % 2].fit(gg if k_ % 2 == 0 else gg2, want_coefficients=False)
                    two_steps(2)
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    two_steps(8)
                    torch.cuda.synchronize()
                    shard_ms2.append((time.perf_counter() - t0_) * 1e3 / 8)
                    del dg, dg2
                bp.set_observations(Y)
                with torch.cuda.stream(streams[1]):
                    handles[1].set_observations(Y2)
                torch.cuda.synchronize()
                sm, se = np.array(shard_ms), np.array(shard_evals, dtype=np.float64)
                out["configs3_emulated"] = {
                    "workload": "BASELINE configs[3]: %d problems = %d shards of %d (contiguous split, varpro_amd/distributed.py"
                                ":shard_range), each shard fitted alone on this one device" % (G * B, G, B),
                    "shards": G, "per_shard_ms_per_step": [float(v) for v in sm],
                    "per_shard_sum_evaluations": shard_evals, "per_shard_max_evaluations": shard_max,
                    "predicted_efficiency": float(sm.mean() / sm.max()),
                    "predicted_efficiency_from_evaluation_counts": float(se.mean() / se.max()),
                    "predicted_fits_per_s_at_%d_gpus" % G: float(G * B / (sm.max() * 1e-3)),
                    "two_batches_in_flight": {
                        "per_shard_ms_per_step": [float(v) for v in shard_ms2],
                        "predicted_efficiency": float(np.mean(shard_ms2) / np.max(shard_ms2)),
                        "predicted_fits_per_s_at_%d_gpus" % G: float(G * B / (np.max(shard_ms2) * 1e-3)),
                        "note": "every shard stepped the way a rank of the real run steps it (step k on handle / stream k mod 2, wall "
                                "clock over 8 steps): the tail of a shard's longest fit overlaps the bulk of the rank's next batch"},
                    "note": "kernel time only (HIP events); the real run adds one RCCL all-reduce of 4 doubles per step",
                }

            # ---- configs[2]: one alpha shared by S = 16384 right-hand sides, m = 2048, triple-exp + offset ----
            S2, m2 = 16384, 2048
            d2 = synth.mrhs_triple_exp(S=S2, m=m2)
            mdl2 = vp.multi_exponential_model(d2["x"], d2["tau_guess"], offset=True)
            Y2 = torch.from_numpy(d2["Y"][None]).to(dev)
            bp2 = vp.BatchProblem(mdl2, Y2, x=torch.from_numpy(d2["x"]).to(dev))
            g2 = torch.from_numpy(d2["tau_guess"][None]).to(dev)
            bp2.set_timing(True)
            ts = []
            for _ in range(6):
                bp2.evaluate(g2, want_residuals=True, want_jacobian=True)
                ts.append(bp2.last_kernel_ms(_lib.VP_KERNEL_EVALUATE))
            ev2_ms = min(ts[1:])
            bytes_ev2 = T * m2 * S2 * (2 + 3)                      # T*m*S*(2+q) = 1.34 GB
            tf, tfe = [], []
            for _ in range(14):  # (the captured graph drops its spare iteration after 8 identical fits: VP_MRHS_EXACT_AFTER)
                t0 = time.perf_counter()
                a2, _C2, rep2 = bp2.fit(g2, want_coefficients=False)
                torch.cuda.synchronize()
                tf.append((time.perf_counter() - t0) * 1e3)
                tfe.append(bp2.last_kernel_ms(_lib.VP_KERNEL_FIT))  # the library's own HIP events around the captured graph
            r2 = bp2.report_to_numpy(rep2)
            fit2_ms = min(tf[1:])
            fit2_event_ms = min(tfe[1:])
            # two global fits in flight: two handles on two HIP streams, one host thread each (the entry point waits on the host
            # for the fit's active count; ctypes releases the GIL) -- a fit is a chain of {51 us pass over y, 18 us step on one
            # wave group}: the passes of one fit run in the step gaps of the other
            bp2.set_timing(False)
            Y2_b = Y2.clone()
            x2_dev = torch.from_numpy(d2["x"]).to(dev)
            ms2_two = in_flight_ms(lambda: vp.BatchProblem(mdl2, Y2_b, x=x2_dev), lambda h_: h_.fit(g2, want_coefficients=False), 2, 12, 14, threads=True)
            ms2_four = in_flight_ms(lambda: vp.BatchProblem(mdl2, Y2_b, x=x2_dev), lambda h_: h_.fit(g2, want_coefficients=False), 4, 12, 14, threads=True)
            del Y2_b
            out["configs2"] = {
                "workload": "BASELINE configs[2]: global fit, 1 alpha shared by %d right-hand sides, m=%d, triple-exp+offset" % (S2, m2),
                "trait_evaluation_ms": ev2_ms, "global_fit_ms": fit2_ms, "global_fit_event_ms": fit2_event_ms,
                "evaluations": int(r2["n_evals"][0]),
                "two_fits_in_flight": {"ms_per_fit": ms2_two, "what": "two handles, two HIP streams, one host thread each; wall clock over 24 fits / 24, best of three passes"},
                "four_fits_in_flight": {"ms_per_fit": ms2_four, "hbm_frac": T * m2 * S2 * int(r2["n_evals"][0]) / (ms2_four * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "what": "four handles: the passes of the other fits fill every step gap; the y re-reads (9 x 268 MB per fit) then run at this fraction of HBM"},
                "termination": int(r2["termination"][0]),
                "max_abs_tau_error": float(np.abs(a2.cpu().numpy()[0] - d2["tau_true"]).max()),
                "roofline": {"kernel": "mrhs_coop_out_kernel (workgroup-cooperative trait-level pass; + mrhs_factor_kernel): y in, r and J out", "bound": "hbm",
                             "achieved": bytes_ev2 / (ev2_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bytes_ev2 / (ev2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": bytes_ev2,
                             "traffic": committed_traffic("mrhs_coop_out_kernel"),
                             "traffic_source": traffic_source("mrhs_coop_out_kernel")},
                "roofline_fit": {"kernel": "mrhs_coop_dma_kernel (MODE 0: workgroup-cooperative, LDS-DMA prefetch): y re-read once per LM "
                                           "evaluation; fraction over the WHOLE global fit (wall clock of vp_fit) incl. mrhs_step_kernel "
                                           "(LM step + factorisation of the next trial point) between the passes; the fit is one captured graph",
                                 "bound": "hbm", "traffic": committed_traffic("mrhs_coop_dma_kernel"),
                                 "traffic_source": traffic_source("mrhs_coop_dma_kernel"),
                                 "achieved": T * m2 * S2 * int(r2["n_evals"][0]) / (fit2_ms * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": T * m2 * S2 * int(r2["n_evals"][0]) / (fit2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            }
            bp2.close()
            del Y2

            # ---- configs[4]: fp32 five exponentials + offset, m = 4096, batch 8192 ----
            B4, m4 = 8192, 4096
            d4 = synth.multi_exp_batch(B4, 5, m4, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05,
                                       dtype=np.float32)
            mdl4 = vp.multi_exponential_model(d4["x"], d4["tau_guess"][0], dtype=np.float32)
            Y4 = torch.from_numpy(d4["Y"]).to(dev)
            g4 = torch.from_numpy(d4["tau_guess"]).to(dev)
            bp4 = vp.BatchProblem(mdl4, Y4, x=torch.from_numpy(d4["x"]).to(dev))
            ms4 = event_ms(lambda: bp4.fit(g4, want_coefficients=False), 5, 2)
            _a4, _c4, rep4 = bp4.fit(g4, want_coefficients=False)
            r4 = bp4.report_to_numpy(rep4)
            x4_dev = torch.from_numpy(d4["x"]).to(dev)
            ms4_two = in_flight_ms(lambda: vp.BatchProblem(mdl4, Y4, x=x4_dev), lambda h_: h_.fit(g4, want_coefficients=False), 2, 5, 1)
            # the fit runs on the fp64 Gram matrix of [Phi | y | dPhi] (vp_fitg.hpp).  ALGORITHMIC flops of one Gram
            # evaluation, as priced since round 2: per row the 66 + 11 inner products of the 11 columns + constant
            # (66 FMAs + 11 adds), 5 exponential + 10 derivative multiplies = 158 fp64 flops.  The kernel EXECUTES far
            # fewer: the moment form needs 133 per row, and on a uniform grid with unit weights (this workload) the 55
            # moments that do not depend on y come from a closed form (doubling recurrence over the bits of m), which
            # left 10 multiplies + 11 FMAs + 1 add = 33 per row through round 4.  Round 5: the 10 y-dependent moments are
            # polynomials in rho_k = e^{-dt/tau_k} -- Horner over a lane's 4 rows, times the lane's anchor: 43 FMAs + 9
            # multiplies per 4 rows = 23.75 flops per row (sum y^2 and sum y once per fit).  Both fractions are reported; the
            # launch is bound by its LONGEST fit (evaluations x {moment pass + lane-serial bookkeeping}), not by the fp64 pipe.
            flops4 = m4 * 158
            flops4_exec = m4 * 95 // 4
            tf4 = float(r4["n_evals"].sum()) * flops4 / (ms4 * 1e-3) / 1e12
            out["configs4"] = {
                "workload": "BASELINE configs[4]: %d fp32 fits, five exponentials + offset (n=6, q=5), m=%d" % (B4, m4),
                "fits_per_s": B4 / (ms4 * 1e-3), "ms_per_step": ms4, "mean_evaluations_per_fit": float(r4["n_evals"].mean()),
                "two_batches_in_flight": {"ms_per_batch": ms4_two, "fits_per_s": B4 / (ms4_two * 1e-3),
                                          "note": "two handles on two HIP streams: the ~1 ms in which a launch runs only its fits "
                                                  "past ~24 evaluations overlaps the next batch's bulk"},
                "fraction_failed": float((r4["termination"] <= 0).mean()),
                "longest_fit_evaluations": int(r4["n_evals"].max()),
                "us_per_round_of_the_longest_fit": ms4 * 1e3 / float(r4["n_evals"].max()),
                "roofline": {"kernel": "fitg2_kernel<5 exp + offset> (fp32 data, fp64 moment/Gram pass with closed-form y-independent "
                                       "moments + Cholesky-based LM; per CU one 8-wave workgroup = 7 streaming waves + 1 bookkeeping "
                                       "wave over a pool of 32 problem slots; passes of long fits split over 4 waves)",
                             "bound": "latency (the longest fit's chain of rounds: evaluations x {moment pass + LM bookkeeping})",
                             "timeline": "tools/cfg4_timeline.py on a -DVP_FITG_TIMELINE=1 build (round 5): the first ~1.0 ms every workgroup holds "
                                         "32..8 live fits and both roles are busy (bookkeeping wave 74 % busy, 6.9 slots per trip of 11.6 us; "
                                         "stream waves 9.4 us per pass or part at 2 waves per SIMD, 10.6 before the y stream ran 8 chunks "
                                         "deep) -- a round costs 41-49 us there; 99 % of the fits are done at 1.25 ms, the rest is the fits "
                                         "past ~32 evaluations at the 22-26 us of a lone round (a split pass, then the dependent fp64 "
                                         "bookkeeping: Cholesky of the 6x6 Gram, pivoted Cholesky of J^T J, lmpar on the Cholesky factor "
                                         "of R^T R + par D^2 -- lmpar_chol)",
                             "achieved": tf4 * flops4_exec / flops4, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf4 * flops4_exec / flops4 / FP64_VALU_PEAK_TFLOPS,
                             "flops_executed_per_evaluation": flops4_exec,
                             "note": "frac = fp64 flops the kernel EXECUTES (23.75 per row: closed-form y-independent moments, Horner form of the others) over the "
                                     "fp64 VALU peak; against the 158 flop/row Gram model of round 2 it would read %.2f" % (tf4 / FP64_VALU_PEAK_TFLOPS),
                             "hbm_bytes_per_evaluation": 4 * m4, "y_reread_bytes_per_launch": 4.0 * m4 * float(r4["n_evals"].sum()),
                             "traffic": committed_traffic("fitg2_kernel"), "traffic_source": traffic_source("fitg2_kernel")},
            }
            cfg4_check = (d4["x"], d4["Y"], _a4.cpu().numpy() if hasattr(_a4, "cpu") else np.asarray(_a4), r4)
            bp4.close()
            del Y4

        if world == 1 and not args.no_side_configs:
            # ---- the S = 1 evaluation boundary: what a trait-level caller of a BATCH pays per LM
            # evaluation -- vp_evaluate with r and J out; B_eval = T [m S (2 + q) + n S + q] = 32808 B per problem ----
            r_ev = torch.empty((B, m), dtype=torch.float64, device=dev)
            J_ev = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
            C_ev = torch.empty((B, 3), dtype=torch.float64, device=dev)
            cost_ev = torch.empty((B,), dtype=torch.float64, device=dev)
            st_ev = torch.empty((B,), dtype=torch.int32, device=dev)
            import ctypes as C_

            def vptr(t):
                return C_.c_void_p(t.data_ptr()) if t is not None else None

            def ev_call():
                _lib.check(bp.lib.vp_evaluate(bp._h, vptr(guess), vptr(r_ev), vptr(J_ev), vptr(C_ev), vptr(cost_ev), vptr(st_ev)))

            def ev_call_r():  # set_params + residuals: what the LM driver asks for at every inner trial point
                _lib.check(bp.lib.vp_evaluate(bp._h, vptr(guess), vptr(r_ev), None, vptr(C_ev), vptr(cost_ev), vptr(st_ev)))

            ev_ms = event_ms_each(ev_call, 20, 3)
            ev_ms_b2b = event_ms(ev_call, 20, 3)
            ev_r_ms = event_ms_each(ev_call_r, 20, 3)
            bytes_ev = B * T * (m * (2 + 2) + 3 + 2)
            bytes_ev_r = B * T * (m * 2 + 3 + 2)
            out["evaluate_boundary"] = {
                "workload": "vp_evaluate, B=%d, m=%d, fp64: alpha in; y read; r, J, c, cost out (the trait-level set_params + "
                            "residuals + jacobian of a batch in one launch)" % (B, m),
                "ms": ev_ms, "bytes_per_problem": T * (m * 4 + 5),
                "residuals_only": {"ms": ev_r_ms, "bytes_per_problem": T * (m * 2 + 5),
                                   "achieved_GBps": bytes_ev_r / (ev_r_ms * 1e-3) / 1e9,
                                   "frac": bytes_ev_r / (ev_r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "what": "y in; r, c, cost out (set_params + residuals: the LM driver's inner trial point)"},
                "roofline": {"kernel": "evaluate_kernel<MODE 2, FULL> (split: [exp | y] factored and r written, then the derivative "
                                       "columns rebuilt, projected and written)", "bound": "hbm", "achieved": bytes_ev / (ev_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_ev / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "bytes_per_launch": bytes_ev, "avg_launch_ms": ev_ms, "back_to_back_ms_per_launch": ev_ms_b2b,
                             "traffic": committed_traffic("evaluate_kernel"),
                             "traffic_source": traffic_source("evaluate_kernel")},
            }

            # ---- a model OUTSIDE the descriptor language (vp_batch_create_external): the caller hands over Phi and dPhi, the
            # device does weighting / QR / solve / residual / Kaufman J.  Columns here: the double-exponential model's own
            # (written once by vp_basis -- a stand-in for the caller's closures, so the result can be compared with vp_evaluate)
            phi_x = torch.empty((B, 3, m), dtype=torch.float64, device=dev)
            dphi_x = torch.empty((B, 2, m), dtype=torch.float64, device=dev)
            bp.basis(guess, skip_invariant=False, out_phi=phi_x, out_dphi=dphi_x)
            bpx = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Y)
            r_x, J_x = torch.empty_like(r_ev), torch.empty_like(J_ev)
            C_x, cost_x, st_x = torch.empty_like(C_ev), torch.empty_like(cost_ev), torch.empty_like(st_ev)

            def x_full():
                _lib.check(bpx.lib.vp_evaluate_with_basis(bpx._h, vptr(guess), vptr(phi_x), vptr(dphi_x), vptr(r_x), vptr(J_x),
                                                          vptr(C_x), vptr(cost_x), vptr(st_x)))

            def x_in():
                _lib.check(bpx.lib.vp_evaluate_with_basis(bpx._h, vptr(guess), vptr(phi_x), None, None, None, vptr(C_x),
                                                          vptr(cost_x), vptr(st_x)))

            xf_ms = event_ms_each(x_full, 20, 3)
            xf_ms_b2b = event_ms(x_full, 20, 3)
            xi_ms = event_ms_each(x_in, 20, 3)
            ev_call()
            x_full()
            torch.cuda.synchronize()
            okx = (st_ev == 0) & (st_x == 0)
            dJ_all = ((J_x - J_ev).abs().amax(dim=(1, 2)) / J_ev.abs().amax(dim=(1, 2)))[okx]
            dJ, dJ_med, dJ_share = float(dJ_all.max()), float(dJ_all.median()), float((dJ_all <= 1e-10).double().mean())
            dr = float(((r_x - r_ev).abs().amax(dim=1) / Y.abs().amax(dim=1))[okx].max())
            dC = float(((C_x - C_ev).abs().amax(dim=1) / C_ev.abs().amax(dim=1))[okx].max())
            bytes_xin = B * T * m * (3 + 1)            # Phi (n) + y
            bytes_xfull = B * T * m * (3 + 2 + 1 + 1 + 2)  # Phi (n) + dPhi (p) + y in, r + J (q) out
            out["external_model"] = {
                "workload": "vp_evaluate_with_basis, B=%d, m=%d, fp64, n=3, q=2, p=2: caller-evaluated Phi / dPhi in (device "
                            "pointers), device weighting + QR / solve + residual + Kaufman J" % (B, m),
                "ms_phi_dphi_in_r_J_out": xf_ms, "ms_phi_in_c_cost_out": xi_ms,
                "max_rel_diff_vs_vp_evaluate": {"c": dC, "r": dr, "J": dJ, "J_median": dJ_med, "J_share_within_1e-10": dJ_share,
                                                "note": "max over the 65 536 problems; the J maximum sits on the ~0.1 % of guesses at which "
                                                        "D_k c is almost inside span(Phi) (the projector cancels most of its input): "
                                                        "tests/test_gpu_eval_census.py holds both handles against the oracle on every "
                                                        "problem and arbitrates those by long double (profiles/r04_eval_census.json)"},
                "roofline": {"kernel": "ext_evaluate_kernel<double, 3, 2, 16, true> (columns loaded, not built)", "bound": "hbm",
                             "achieved": bytes_xfull / (xf_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bytes_xfull / (xf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": bytes_xfull,
                             "avg_launch_ms": xf_ms, "back_to_back_ms_per_launch": xf_ms_b2b,
                             "input_stream_GBps": B * T * m * (3 + 2 + 1) / (xf_ms * 1e-3) / 1e9,
                             "traffic": committed_traffic("ext_evaluate_kernel"), "traffic_source": traffic_source("ext_evaluate_kernel")},
                "roofline_set_params": {"kernel": "ext_evaluate_kernel<double, 3, 1, 16, false> (Phi and y in; c, cost out)", "bound": "hbm",
                                        "achieved": bytes_xin / (xi_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": bytes_xin / (xi_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": bytes_xin},
            }
            # ---- the same boundary for LONG problems and fp32 (round 5): W waves of a workgroup share the columns of one problem
            # (ext_evaluate_kernel<.., W>): one pass over the caller's arrays up to 4 096 rows (four waves), 8 192 fp64 /
            # 16 384 fp32 (eight); beyond that the streamed kernel of vp_blk_ext.hpp (any m, two passes over the columns)
            long_legs = {}
            for tdt, ml, Bl, kern, tkey in ((torch.float32, 4096, 32768, "ext_evaluate_kernel<float, 3, 2, 16, true, 4> (four waves per problem)", "ext_evaluate_kernel_w4"),
                                            (torch.float64, 8192, 8192, "ext_evaluate_kernel<double, 3, 2, 16, true, 8> (eight waves per problem)", "ext_evaluate_kernel_w8"),
                                            (torch.float64, 10000, 4096, "blk::ext_stream_evaluate_kernel<double, 3, 2, 8> (rows streamed in blocks, two passes over the caller's columns: 15 column transfers for 9 algorithmic; round 6: lane-private pass 1 + row-local pass 2 for well-conditioned problems)", "ext_stream_evaluate_kernel")):
                Tl = 4 if tdt == torch.float32 else 8
                gl_ = torch.Generator(device=dev)
                gl_.manual_seed(0x5EED77)
                xl = torch.linspace(0.0, 12.5, ml, dtype=torch.float64, device=dev)
                t1 = 0.5 + 1.5 * torch.rand((Bl, 1), generator=gl_, dtype=torch.float64, device=dev)
                t2 = 2.5 + 5.5 * torch.rand((Bl, 1), generator=gl_, dtype=torch.float64, device=dev)
                cl = 1.0 + 99.0 * torch.rand((Bl, 3), generator=gl_, dtype=torch.float64, device=dev)
                phl = torch.empty((Bl, 3, ml), dtype=tdt, device=dev)
                dpl = torch.empty((Bl, 2, ml), dtype=tdt, device=dev)
                e1_, e2_ = torch.exp(-xl[None] / t1), torch.exp(-xl[None] / t2)
                phl[:, 0], phl[:, 1], phl[:, 2] = e1_.to(tdt), e2_.to(tdt), 1.0
                dpl[:, 0], dpl[:, 1] = (e1_ * xl[None] / (t1 * t1)).to(tdt), (e2_ * xl[None] / (t2 * t2)).to(tdt)
                Yl = (cl[:, 0:1] * e1_ + cl[:, 1:2] * e2_ + cl[:, 2:3])
                Yl = (Yl + args.noise * Yl.abs().amax(dim=1, keepdim=True) * torch.randn(Yl.shape, generator=gl_, dtype=torch.float64, device=dev)).to(tdt)
                del e1_, e2_
                al = torch.cat([t1, t2], 1).to(tdt) * 1.1
                bpl = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)], dtype=np.float32 if tdt == torch.float32 else np.float64), Yl)
                rl, Jl = torch.empty((Bl, ml), dtype=tdt, device=dev), torch.empty((Bl, 2, ml), dtype=tdt, device=dev)
                Cl, costl, stl = torch.empty((Bl, 3), dtype=tdt, device=dev), torch.empty((Bl,), dtype=torch.float64, device=dev), torch.empty((Bl,), dtype=torch.int32, device=dev)

                def xl_full():
                    _lib.check(bpl.lib.vp_evaluate_with_basis(bpl._h, vptr(al), vptr(phl), vptr(dpl), vptr(rl), vptr(Jl), vptr(Cl), vptr(costl), vptr(stl)))

                xl_ms = event_ms_each(xl_full, 8, 2)
                torch.cuda.synchronize()
                byl = Bl * Tl * ml * 9
                long_legs["%s_m%d" % ("f32" if Tl == 4 else "f64", ml)] = {
                    "workload": "vp_evaluate_with_basis, B=%d, m=%d, %s, n=3, q=2, p=2 (Phi, dPhi, y in; r, J out)" % (Bl, ml, "fp32" if Tl == 4 else "fp64"),
                    "ms": xl_ms, "status_ok_share": float((stl == 0).double().mean()),
                    "roofline": {"kernel": kern, "bound": "hbm", "achieved": byl / (xl_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": byl / (xl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": byl, "avg_launch_ms": xl_ms,
                                 "traffic": committed_traffic(tkey), "traffic_source": traffic_source(tkey)}}
                bpl.close()
                del phl, dpl, Yl, rl, Jl
            out["external_model"]["long_problems"] = long_legs

            # ---- the FIT of such a batch by reverse communication (vp_fit_begin / vp_fit_step_with_basis / vp_fit_end): the LM
            # driver of every problem on the device, the model with the caller.  The caller here is the double-exponential model
            # evaluated ON THE DEVICE by the caller's own kernels (vp_basis of the descriptor handle while most problems are
            # active, a torch expression over the problems that still want columns in the tail) -- the headline workload, so the
            # result can be held against vp_fit of the same problems
            del r_x, J_x, r_ev, J_ev
            xt_row = x[None, None, :]

            act_idx = torch.empty((B,), dtype=torch.int32, device=dev)
            act_cnt = torch.empty((1,), dtype=torch.int32, device=dev)

            def caller_model(alpha, want, n_active):
                if want is None or n_active * 4 >= B:
                    bp.basis(alpha, skip_invariant=False, out_phi=phi_x, out_dphi=dphi_x)
                else:  # the tail: only the problems still running -- the handle's compacted active set (vp_fit_active_set:
                    # two asynchronous device copies, no scan of want[] and no host synchronisation)
                    bpx.fit_active_set(act_idx, act_cnt)
                    idx = act_idx[:n_active].long()
                    a_ = alpha[idx][:, :, None]
                    e_ = torch.exp(-xt_row / a_)
                    phi_x[idx, 0:2] = e_
                    dphi_x[idx] = e_ * xt_row / (a_ * a_)

            def stepped_fit(step_events=None):
                bpx.fit_begin(guess)
                alpha_, want_, nact, steps_ = guess, None, B, 0
                while nact > 0 and steps_ < 400:
                    caller_model(alpha_, want_, nact)
                    if step_events is not None:
                        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0_.record()
                    # the active count is read back (a host synchronisation) every fourth step while many problems run: the
                    # count only shrinks, a stale one makes the caller evaluate a superset
                    look_ = (steps_ + 1) % 4 == 0 or nact < 64
                    alpha_, want_, na_ = bpx.fit_step_with_basis(phi_x, dphi_x, want_count=look_)
                    nact = na_ if look_ else nact
                    if step_events is not None:
                        e1_.record()
                        step_events.append((e0_, e1_, nact))
                    steps_ += 1
                return bpx.fit_end(want_coefficients=False) + (steps_,)

            stepped_fit()
            torch.cuda.synchronize()
            sev = []
            t0x = time.perf_counter()
            a_xf, _cxf, rep_xf, steps_xf = stepped_fit(sev)
            torch.cuda.synchronize()
            xfit_s = time.perf_counter() - t0x
            rxf = bpx.report_to_numpy(rep_xf)
            a_vf, _cvf, rep_vf = bp.fit(guess, want_coefficients=False)
            rvf = bp.report_to_numpy(rep_vf)
            both_ok = (rxf["termination"] > 0) & (rvf["termination"] > 0)
            rel_obj = np.abs(rxf["objective"] - rvf["objective"])[both_ok] / rvf["objective"][both_ok]
            step0_ms = sev[0][0].elapsed_time(sev[0][1])
            bytes_step = B * T * m * (3 + 2 + 1)  # Phi (n) + dPhi (p) + y of every (active) problem; out: alpha_trial + want
            out["external_fit"] = {
                "workload": "vp_fit_begin / vp_fit_step_with_basis / vp_fit_end, B=%d, m=%d, fp64, n=3, q=2, p=2 (the headline problems as a "
                            "CALLER-EVALUATED model: columns written by the caller's kernels on the device, LM drivers on the device)" % (B, m),
                "fits_per_s": B / xfit_s, "ms_per_fit_of_the_batch": xfit_s * 1e3, "steps": steps_xf,
                # the library's share: HIP events around every vp_fit_step_with_basis (evaluation + LM launches, the 4-byte count
                # read-back every fourth step); the rest of the wall clock is the CALLER's model (here: vp_basis of a descriptor
                # handle while most problems run, eight torch launches over the compacted active set in the tail)
                "ms_inside_the_steps": sum(a_.elapsed_time(b_) for a_, b_, _n in sev),
                "fits_per_s_steps_only": B / (sum(a_.elapsed_time(b_) for a_, b_, _n in sev) * 1e-3),
                "mean_evaluations_per_fit": float(rxf["n_evals"].mean()),
                "bytes_crossing_the_boundary_per_iteration": {"out_alpha_trial_and_want": B * (2 * T + 4),
                                                              "in_columns_by_device_pointer": bytes_step - B * T * m},
                "vs_vp_fit_of_the_same_problems": {"same_success_class": float(((rxf["termination"] > 0) == (rvf["termination"] > 0)).mean()),
                                                   "objective_rel_diff_median": float(np.median(rel_obj)), "objective_rel_diff_max": float(rel_obj.max()),
                                                   "evaluations": [int(rxf["n_evals"].sum()), int(rvf["n_evals"].sum())]},
                "roofline": {"kernel": "ext_fit_eval_kernel<double, 3, 2, 2, 16, 1> + ext_fit_lm_kernel<double, 2> (first step: every problem active; events around both launches)", "bound": "hbm",
                             "achieved": bytes_step / (step0_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bytes_step / (step0_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": bytes_step,
                             "avg_launch_ms": step0_ms, "traffic": committed_traffic("ext_fit_eval_kernel"),
                             "traffic_source": traffic_source("ext_fit_eval_kernel")},
            }
            bpx.close()
            del phi_x, dphi_x

        if world == 1 and not args.no_side_configs:
            # ---- the LENGTH-AGNOSTIC kernels (vp_block.hpp): rows streamed in blocks through a TSQR update of an
            # (n+1+p)^2 triangle -- what a problem lands on beyond the largest register-resident set (2048 rows) ----
            legs = {}
            for ms, Bs in ((10000, 16384), (100000, 2048)):
                ds = synth.double_exp_batch(Bs, m=ms, noise=args.noise)
                mdls = vp.multi_exponential_model(ds["x"], ds["tau_guess"][0])
                bps = vp.BatchProblem(mdls, torch.from_numpy(ds["Y"]).to(dev), x=torch.from_numpy(ds["x"]).to(dev))
                gs = torch.from_numpy(ds["tau_guess"]).to(dev)
                mss = event_ms(lambda: bps.fit(gs, want_coefficients=False), 3, 1)
                _as, _cs, reps = bps.fit(gs, want_coefficients=False)
                rs = bps.report_to_numpy(reps)
                evs = float(rs["n_evals"].sum())
                fl = ms * 2 * 28 + 2 * ms * 9 + 4 * ms * 3 + (4 * 3 + 2) * ms * 2   # SURVEY 8(d) at this m
                legs["m%d" % ms] = {
                    "workload": "B=%d double-exp+offset fits, m=%d, fp64 (no resident kernel set is this long)" % (Bs, ms),
                    "fits_per_s": Bs / (mss * 1e-3), "ms_per_step": mss, "mean_evaluations_per_fit": evs / Bs,
                    "fraction_failed": float((rs["termination"] <= 0).mean()),
                    "roofline": {"kernel": "blk_fit_kernel<double, MultiExpModel<2, true>, 20, false, W, TC> (blocks of 1 280 rows, one wave per SIMD, grid values computed: the LDS ring holds the data alone; four waves per problem at m = 100 000)", "bound": "fp64_valu",
                                 "achieved": evs * fl / (mss * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": evs * fl / (mss * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                 "y_stream_GBps": evs * ms * T / (mss * 1e-3) / 1e9, "algorithmic_bytes_per_launch": evs * ms * T,
                                 "traffic": committed_traffic_grid("blk_fit_kernel", Bs * 64) or committed_traffic_grid("blk_fit_kernel", Bs * 256),  # (launches smaller than the device: four waves per problem)
                                 "traffic_source": traffic_source("blk_fit_kernel"),
                                 "note": "y (8 m bytes) is re-read per evaluation through the LDS ring; the grid comes from L2"},
                }
                if ms == 10000:
                    # the same long problems as a CALLER-EVALUATED model fitted by reverse communication: beyond 4 096 rows the
                    # step streams the caller's columns in row blocks (vp_blk_extfit.hpp); columns by a torch expression on the
                    # device for every problem and step (the simplest caller)
                    Bx = 4096
                    Yx_, gx_ = torch.from_numpy(ds["Y"][:Bx]).to(dev), gs[:Bx].contiguous()
                    xrow = torch.from_numpy(ds["x"]).to(dev)[None, None, :]
                    bxs = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Yx_)
                    phis = torch.ones((Bx, 3, ms), dtype=torch.float64, device=dev)
                    dphis = torch.empty((Bx, 2, ms), dtype=torch.float64, device=dev)

                    def long_stepped(evts=None):
                        bxs.fit_begin(gx_)
                        al_, nact_, st_ = gx_, Bx, 0
                        while nact_ > 0 and st_ < 400:
                            a3 = al_[:, :, None]
                            torch.exp(-xrow / a3, out=phis[:, 0:2])
                            torch.mul(phis[:, 0:2], xrow / (a3 * a3), out=dphis)
                            if evts is not None and st_ == 0:
                                e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                e0_.record()
                            al_, _w, nact_ = bxs.fit_step_with_basis(phis, dphis)
                            if evts is not None and st_ == 0:
                                e1_.record()
                                evts.append((e0_, e1_))
                            st_ += 1
                        return bxs.fit_end(want_coefficients=False) + (st_,)

                    long_stepped()
                    torch.cuda.synchronize()
                    evx = []
                    t0s = time.perf_counter()
                    _ax, _cx, repx, stx = long_stepped(evx)
                    torch.cuda.synchronize()
                    xs_s = time.perf_counter() - t0s
                    rxs = bxs.report_to_numpy(repx)
                    okb = (rxs["termination"] > 0) & (rs["termination"][:Bx] > 0)
                    relx = np.abs(rxs["objective"] - rs["objective"][:Bx])[okb] / rs["objective"][:Bx][okb]
                    st0 = evx[0][0].elapsed_time(evx[0][1])
                    byx = Bx * T * ms * 6
                    legs["m%d" % ms]["as_caller_evaluated_model"] = {
                        "workload": "vp_fit_begin / vp_fit_step_with_basis / vp_fit_end, B=%d of the same problems, m=%d: columns by a torch expression on the device every step" % (Bx, ms),
                        "fits_per_s_including_the_callers_columns": Bx / xs_s, "steps": stx, "mean_evaluations_per_fit": float(rxs["n_evals"].mean()),
                        "vs_vp_fit_of_the_same_problems": {"same_success_class": float(((rxs["termination"] > 0) == (rs["termination"][:Bx] > 0)).mean()),
                                                           "objective_rel_diff_median": float(np.median(relx)), "objective_rel_diff_max": float(relx.max())},
                        "roofline": {"kernel": "blk::ext_fit_stream_eval_kernel<double, 3, 2, 2, 4> + ext_fit_lm_kernel<double, 2> (first step: every problem active)", "bound": "hbm",
                                     "achieved": byx / (st0 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byx / (st0 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "bytes_per_launch": byx, "avg_launch_ms": st0, "traffic": committed_traffic("ext_fit_stream_eval_kernel"),
                                     "traffic_source": traffic_source("ext_fit_stream_eval_kernel")}}
                    bxs.close()
                    del phis, dphis, Yx_
                bps.close()
                del ds, gs
            out["streamed_rows"] = legs

        # ---- generic fallback kernels (vp_generic.hpp): a model / size WITHOUT a specialised kernel set ----
        if world == 1 and not args.no_side_configs:
            # the O'Leary-Rust example model (shared_test_code/src/models.rs:397-425: exp(-a2 t) cos(a3 t), exp(-a1 t) cos(a2 t);
            # n = 2, q = 3, 4 dependency pairs, a shared parameter) at m = 5000 -- beyond the 4096 rows of its largest specialised
            # set (round 3 added a 4-wave set for 1024 < m <= 4096: m = 3000, this leg's length until then, no longer falls back)
            Bg, mg = 4096, 5000
            tg = np.linspace(0.0, 1.5, mg)
            rg = synth.SplitMix64(np.uint64(0x5EED3000) + np.arange(Bg, dtype=np.uint64))
            at = np.stack([1.0 * (1 + 0.1 * rg.uniform(-1, 1)), 2.5 * (1 + 0.1 * rg.uniform(-1, 1)), 4.0 * (1 + 0.1 * rg.uniform(-1, 1))], 1)
            cg = np.stack([rg.uniform(4.0, 8.0), rg.uniform(0.5, 2.0)], 1)
            Yg = (cg[:, :1] * np.exp(-at[:, 1:2] * tg[None]) * np.cos(at[:, 2:3] * tg[None])
                  + cg[:, 1:2] * np.exp(-at[:, 0:1] * tg[None]) * np.cos(at[:, 1:2] * tg[None]))
            Yg = Yg + 1e-3 * np.abs(Yg).max(1, keepdims=True) * rg.normal(mg)
            gg0 = at * np.stack([1 + 0.1 * rg.uniform(-1, 1) for _ in range(3)], 1)
            mdlg = (vp.SeparableModelBuilder(["alpha1", "alpha2", "alpha3"]).initial_parameters(gg0[0]).independent_variable(tg)
                    .function(["alpha2", "alpha3"], vp.basis.EXP_COS).partial_deriv("alpha2").partial_deriv("alpha3")
                    .function(["alpha1", "alpha2"], vp.basis.EXP_COS).partial_deriv("alpha1").partial_deriv("alpha2").build())
            bpg = vp.BatchProblem(mdlg, torch.from_numpy(Yg).to(dev), x=torch.from_numpy(tg).to(dev))
            ggd = torch.from_numpy(gg0).to(dev)
            msg = event_ms(lambda: bpg.fit(ggd, want_coefficients=False), 3, 1)
            _ag, _cg2, repg = bpg.fit(ggd, want_coefficients=False)
            rgp = bpg.report_to_numpy(repg)
            evg = float(rgp["n_evals"].sum())
            out["generic_fallback"] = {
                "workload": "O'Leary exp*cos model (n=2, q=3, p=4, shared parameter), B=%d, m=%d, fp64: no register-resident kernel "
                            "set at this m -> round 4: the length-agnostic blk_fit_kernel<RtModel<2,3,4>> (round 3: gen_fit_kernel, "
                            "one workgroup per problem with its columns in a global-memory workspace, 0.042 M fits/s)" % (Bg, mg),
                "fits_per_s": Bg / (msg * 1e-3), "ms_per_step": msg, "mean_evaluations_per_fit": evg / Bg,
                "fraction_failed": float((rgp["termination"] <= 0).mean()),
                "roofline": {"kernel": "blk_fit_kernel<double, RtModel<2, 3, 4>, 8>", "bound": "fp64_valu (column build: exp, cos, sin per element "
                                                                                          "of two basis functions and four derivative columns)",
                             "y_stream_GBps": evg * mg * 8.0 / (msg * 1e-3) / 1e9,
                             "traffic": committed_traffic("blk_fit_kernel_rt"), "traffic_source": traffic_source("blk_fit_kernel_rt"),
                             "algorithmic_bytes_per_launch": evg * mg * 8.0,
                             "note": "one wavefront per problem, one wave per SIMD (the run-time-descriptor column build needs the registers); "
                                     "y re-read per evaluation"},
            }
            bpg.close()

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores, rank 0, N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cpu_model, sockets, phys_cores, usable, quota = CPU_TOPOLOGY
        # one pinned thread per physical core, capped by the container's CPU bandwidth quota (a cgroup limit of N CPUs
        # throttles anything beyond N runnable threads: measured on the GPU box, tools/cpu_scaling_probe.py)
        threads = max(1, phys_cores if quota is None else min(phys_cores, int(quota)))
        # (i) SURVEY 8(d)(i): single thread on configs[0] -- the reference is single-threaded by design
        c0 = synth.config0()
        mdl0 = vp.multi_exponential_model(c0["x"], c0["tau_guess"])
        reps0 = 200
        Y0 = np.tile(c0["y"][None, :], (reps0, 1))
        G0 = np.tile(c0["tau_guess"][None, :], (reps0, 1))
        _a, _c, rep0, secs0 = O.fit_batch(mdl0, c0["x"], Y0, G0, n_threads=1)
        # (ii) single thread on the batch workload (the yardstick for the parallel efficiency)
        n1 = min(B, 6144)  # >= 2 s of single-thread work: a yardstick long enough that parallel_efficiency <= 1
        t1 = time.perf_counter()
        _a, _c, rep1, secs1 = O.fit_batch(mdl, d["x"], d["Y"][:n1], d["tau_guess"][:n1], n_threads=1)
        rate1 = n1 / secs1
        # (iii) all physical cores, pinned, static split; sample sized for ~cpu_seconds of wall time
        pilot_n = min(B, 32 * threads)
        t1 = time.perf_counter()
        O.fit_batch(mdl, d["x"], d["Y"][:pilot_n], d["tau_guess"][:pilot_n], n_threads=threads)
        pilot = time.perf_counter() - t1
        n_cpu = int(min(max(pilot_n, args.cpu_seconds * pilot_n / max(pilot, 1e-6)), 262144))
        dd = d if n_cpu <= B else synth.double_exp_batch(n_cpu, m=m, noise=args.noise)
        # measured TWICE, the faster run is the baseline (the slower one is reported next to it: on a shared host the spread
        # between two boxes of the same CPU model was 30 %, between two runs on one box it is a few percent)
        walls = []
        for _rep in range(2):
            t1 = time.perf_counter()
            a_cpu, _c, rep_cpu, secs_fit_ = O.fit_batch(mdl, dd["x"], dd["Y"][:n_cpu], dd["tau_guess"][:n_cpu], n_threads=threads)
            walls.append((time.perf_counter() - t1, secs_fit_))
        wall, secs_fit = min(walls)
        # parity census: the oracle's fits of this sample against the device's fits of the SAME problems, problem by
        # problem (oracle/census.py; tests/test_gpu_census.py asserts the contract on the whole 65536-problem shard)
        from oracle import census as CS
        n_cen = min(n_cpu, B)
        a_dev, _cd, rep_dev = bp.fit(guess, want_coefficients=False)
        torch.cuda.synchronize()
        cen = CS.census(bp.report_to_numpy(rep_dev)[:n_cen], a_dev.cpu().numpy()[:n_cen], rep_cpu[:n_cen], a_cpu[:n_cen], max_listed=10)
        out["parity_census"] = dict(cen, what="vp_fit vs the oracle on the first %d problems of the timed workload: success class, "
                                              "termination codes, objective, evaluation counts per problem" % n_cen)
        if cfg4_check is not None and "configs4" in out:
            # configs[4]: the objective every successful fit REPORTS against the fp64 oracle's cost at the parameters it RETURNS
            # (thin-SVD solve on the lattice the kernel defines the uniform grid as; tools/cfg4_resolution_probe.py)
            x4c, Y4c, a4c, r4c = cfg4_check
            ok4 = r4c["termination"] > 0
            t04 = float(x4c[0])
            grid4 = t04 + np.arange(x4c.shape[-1]) * ((float(x4c[-1]) - t04) / (x4c.shape[-1] - 1))
            ref4 = O.evaluate_batch(vp.multi_exponential_model(grid4, a4c[0].astype(np.float64)), grid4, Y4c[ok4].astype(np.float64),
                                    a4c[ok4].astype(np.float64), n_threads=threads, want_jac=False)
            rel4 = np.abs(r4c["objective"][ok4] - ref4["cost"]) / ref4["cost"]
            out["configs4"]["reported_objective_vs_oracle_cost_at_the_returned_point"] = {
                "successes": int(ok4.sum()), "median": float(np.median(rel4)), "p99": float(np.percentile(rel4, 99)), "max": float(rel4.max()),
                "share_above_1e-3": float((rel4 > 1e-3).mean()), "share_above_1e-2": float((rel4 > 1e-2).mean()),
                "note": "fp64 Gram evaluation of fp32 data: exact to kappa(Phi)^2 x 1e-13; columns whose Cholesky pivot is below "
                        "1e-10 A_ii are dropped (vp_fitg.hpp gram_phase)"}
        out["cpu_baseline"] = {
            "value": n_cpu / wall, "unit": "fits/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "sockets": sockets, "physical_cores": phys_cores, "logical_cpus_usable": usable,
            "cgroup_cpu_quota": quota,
            "threads": threads, "pinning": "OMP_PLACES=%s OMP_PROC_BIND=%s" % (os.environ.get("OMP_PLACES"), os.environ.get("OMP_PROC_BIND")),
            "sample": "first %d problems of the same synthetic workload, %d pinned OpenMP threads (one per physical core%s), "
                      "%.1f s wall (problem construction + initial evaluation included)"
                      % (n_cpu, threads, "" if quota is None or quota >= phys_cores else
                         ", capped by the container's cgroup CPU quota of %g CPUs" % quota, wall),
            "fits_per_s_inside_fits": n_cpu / max(secs_fit, 1e-9),
            "both_runs_fits_per_s": [n_cpu / w_ for w_, _s in walls],
            "mean_evaluations_per_fit": float(rep_cpu["n_evals"].mean()),
            "single_thread_fits_per_s": rate1, "single_thread_sample": "%d problems, %.2f s" % (n1, secs1),
            "parallel_efficiency": (n_cpu / max(secs_fit, 1e-9)) / (threads * rate1),
            "configs0_single_thread": {"us_per_fit": secs0 / reps0 * 1e6, "evaluations_per_fit": float(rep0["n_evals"].mean()),
                                       "us_per_evaluation": secs0 / reps0 * 1e6 / float(rep0["n_evals"].mean())},
            "note": "CPU restatement of the reference algorithm (oracle/varpro_oracle.c), not the Rust reference",
        }
        # what the whole host would deliver at the measured per-core rate and parallel efficiency, had the container all
        # of its physical cores (an extrapolation, labelled as such; `value` above is what was measured)
        eff = out["cpu_baseline"]["parallel_efficiency"]
        out["cpu_baseline"]["extrapolated_all_physical_cores_fits_per_s"] = rate1 * phys_cores * min(1.0, eff)
        out["cpu_baseline"]["extrapolated_single_socket_fits_per_s"] = rate1 * (phys_cores / sockets) * min(1.0, eff)
        # the single-thread rate is stable to 1 % across boxes, the all-core run is not (shared hosts: 0.67 .. 0.98 parallel
        # efficiency on the same CPU model and quota): cores x single-thread rate is the reproducible yardstick
        out["cpu_baseline"]["cores_x_single_thread_fits_per_s"] = rate1 * threads
        out["gpu_over_cores_x_single_thread"] = value / (rate1 * threads)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        out["gpu_over_cpu_single_socket_extrapolated"] = value / out["cpu_baseline"]["extrapolated_single_socket_fits_per_s"]
    if rank == 0:
        try:
            out["build"] = {"library_bytes": os.path.getsize(_lib.LIB_PATH)}
            out["build"].update(json.load(open(os.path.join(ROOT, "varpro_amd", "lib", "build_info.json"))))
        except Exception:
            pass
        line = emit(out)
        print(line, flush=True)
    for h_ in handles:
        h_.close()
    if use_dist:
        barrier()  # the other ranks wait here while rank 0 takes the per-kernel measurements / CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
