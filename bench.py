"""bench.py -- integrated latent frames/s of the Neural-ODE hot path on MI355X.

One "step" = one `DiffEqSolver.forward(z0, t)` over one batch of synthetic Moving-MNIST-shaped latents
(BASELINE.json configs[1]: ODEConvGRU latents (B,64,16,16), 10 output frames = 9 rk4(3/8) intervals, fp32).
Inputs are resident in HBM before the timed region.  With --gpus N every rank integrates its own batch
(weak scaling, no data-path collective: samples are independent under a fixed-grid solver).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel (wino_persist_kernel: the whole trajectory, Winograd F(2x2,3x3) on exact-fp32 MFMA, in one
                  launch; conv3x3_wino_kernel<4> per layer when the persistent path is off): algorithmic FLOP per launch /
                  average launch duration measured with HIP events over the timed region;
  cpu_baseline -- the oracle (CPU restatement of torchdiffeq 0.2.1 on torch-CPU convs) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide); the bf16 conv is NOT bound by it (launch boundary, staging, epilogue)


def profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate --pmc passes,
    FETCH_SIZE doubled per the gfx950 correction); None if the summary is absent.  Not measured live."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_rocprof_summary.json")) as fh:
            return float(json.load(fh)["notes"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE configs[1]: 64)")
    p.add_argument("--frames", type=int, default=10, help="output time points (10 -> 9 intervals)")
    p.add_argument("--method", default="rk4")
    p.add_argument("--train", action="store_true", help="time forward + backward (gradients w.r.t. z0 and all weights)")
    p.add_argument("--adjoint", action="store_true", help="--train through odeint_adjoint (dopri5: seminorm) instead of backward through the solver")
    p.add_argument("--global-step-control", action="store_true",
                   help="dopri5 on N > 1 ranks: one error norm over the global batch (an all-reduce per attempted step)")
    p.add_argument("--shape", default="A", choices=["A", "V"], help="dynamics: A = ODEConvGRU (headline), V = VidODE latent shape")
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                   help="compute dtype of the 3x3 convs: f32 (headline, exact) or bf16 operands + fp32 accumulate/state (configs[4])")
    p.add_argument("--graph", action="store_true", help="diagnostic: replay the forward trajectory from a captured HIP graph")
    p.add_argument("--side-stream", action="store_true", help="diagnostic: run the timed region on a non-default stream")
    p.add_argument("--adjoint-norm", default="seminorm", choices=["seminorm", "mixed"],
                   help="dopri5 adjoint: seminorm, or torchdiffeq's default mixed norm (parameter block steers the steps)")
    p.add_argument("--rtol", type=float, default=None, help="dopri5 tolerances (default: DiffEqSolver's 1e-4 / 1e-5)")
    p.add_argument("--atol", type=float, default=None)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    return p.parse_args()


def conv_flops(f_channels, batch):
    return sum(2 * batch * ci * co * 9 * 256 for ci, co in zip(f_channels[:-1], f_channels[1:]))


def cpu_baseline(state, z0, t, method, budget_s):
    """Time the oracle on the host cores: whole trajectories of the same workload until ~budget_s.
    torch-CPU convs on 16x16 maps scale badly past a few dozen threads, so a short sweep picks the
    fastest intra-op thread count first (that count is what `cores` reports)."""
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    ws, bs = rm.split_convnet_state(state, "gradient_net.")
    f = rm.ode_func(ws, bs)
    frames = z0.shape[0] * len(t)
    hw = torch.get_num_threads()
    best_n, best_t = hw, float("inf")
    with torch.no_grad():
        for n in sorted({hw, 64, 32, 16, 8}, reverse=True):
            if n > hw:
                continue
            torch.set_num_threads(n)
            torchdiffeq_ref.odeint(f, z0, t, method=method)  # warm-up
            t0 = time.perf_counter()
            torchdiffeq_ref.odeint(f, z0, t, method=method)
            el = time.perf_counter() - t0
            if el < best_t:
                best_n, best_t = n, el
        torch.set_num_threads(best_n)
        n, t0 = 0, time.perf_counter()
        while True:
            torchdiffeq_ref.odeint(f, z0, t, method=method)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 400:
                break
    torch.set_num_threads(hw)
    return {"value": frames * n / el, "unit": "latent frames/s", "cores": best_n, "kind": "port",
            "sample": f"{n} full trajectories of the same workload (B={z0.shape[0]}, T={len(t)}, {method}), "
                      f"{el:.1f} s of torch-CPU fp32, no_grad, {best_n} intra-op threads (fastest of a sweep up to {hw})"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("ODEHIP_BENCH_REHEARSAL"):   # N ranks on ONE GPU over gloo: exercises this file's N > 1 path on a 1-GPU box
            local = 0
            # the persistent trajectory kernel holds every CU and assumes its process owns the GPU (one process per GPU): two of
            # them sharing a card could each hold half the CUs waiting for the other half
            os.environ["ODEHIP_PERSISTENT"] = "0"
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import ode_rl_amd
    if a.dtype == "bf16":
        ode_rl_amd.set_compute_dtype("bf16")
    torch.manual_seed(0)
    # A = ODEConvGRU dynamics (5 x conv 64 -> 64); V = VidODE dynamics (128 -> 64 -> 64 -> 128; BASELINE configs[3])
    C0, chans = (64, [64] * 6) if a.shape == "A" else (128, [128, 64, 64, 128])
    f = ode_rl_amd.ODEFunc(n_inputs=C0, n_outputs=C0, n_layers=3 if a.shape == "A" else 2, n_units=64, downsize=False, nonlinear="relu",
                           final_act=False)
    state = {k: v.detach().clone() for k, v in f.state_dict().items()}
    f = f.to(dev)
    solver = ode_rl_amd.DiffEqSolver(f, a.method, device=dev)
    if a.rtol is not None:
        solver.odeint_rtol = a.rtol
    if a.atol is not None:
        solver.odeint_atol = a.atol
    g = torch.Generator().manual_seed(1234 + rank)
    z0_cpu = torch.randn(a.batch, C0, 16, 16, generator=g) * 0.5
    z0 = z0_cpu.to(dev)
    T = a.frames
    t_cpu = torch.arange(T, 2 * T, dtype=torch.float64) / (2 * T)
    t = t_cpu.to(dev)

    if dist is not None and a.method == "dopri5" and a.global_step_control:
        from ode_rl_amd.dist import enable_global_step_control
        enable_global_step_control(dev)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    gout = torch.randn(T, a.batch, C0, 16, 16, generator=torch.Generator().manual_seed(99)).to(dev) if a.train else None

    def step():
        if not a.train:
            with torch.no_grad():
                return solver(z0, t)
        zz = z0.detach().requires_grad_(True)
        f.zero_grad(set_to_none=False)
        if a.adjoint:
            o = ode_rl_amd.odeint_adjoint(f, zz, t, rtol=solver.odeint_rtol, atol=solver.odeint_atol, method=a.method,
                                          adjoint_options={"norm": a.adjoint_norm} if a.method == "dopri5" else None)
        else:
            o = solver(zz, t)
        o.backward(gout)
        if dist is not None:   # the ONE collective of a training step: flattened-bucket all-reduce of the gradients
            from ode_rl_amd.dist import allreduce_gradients
            allreduce_gradients(f.parameters())
        return o.detach()

    if a.side_stream:   # diagnostic: run on a non-default stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)
    if a.graph and not a.train:   # diagnostic: capture one forward trajectory in a HIP graph and replay it
        gs = torch.cuda.Stream()
        gs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(gs):
            step()
        torch.cuda.current_stream().wait_stream(gs)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = step()
        eager_step = step

        def step():
            graph.replay()
            return graph_out
    for _ in range(a.warmup):
        out = step()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    persist0 = ode_rl_amd._lib.load().odehip_persistent_trajectory_launches()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        out = step()
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    assert out.shape == (T, a.batch, C0, 16, 16) and bool(torch.isfinite(out).all())
    dev_ms = ev0.elapsed_time(ev1)

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    if rank == 0:
        if a.method == "dopri5":
            nfe_per_step = int(ode_rl_amd.last_stats.get("nfe", 0))
            adj = dict(ode_rl_amd.last_adjoint_stats) if (a.train and a.adjoint) else None
        else:
            nfe_per_step = {"rk4": 4, "midpoint": 2, "euler": 1}[a.method] * (T - 1)
        n_convs = len(chans) - 1
        launches = nfe_per_step * n_convs * a.steps * (2 if a.train else 1)   # train: + the dgrad conv of every layer
        if a.method == "dopri5" and a.train and a.adjoint:  # adaptive adjoint: each augmented evaluation = forward + dgrad convs
            launches = (nfe_per_step + 2 * adj.get("nfe", 0)) * n_convs * a.steps
        elif a.method == "dopri5" and a.train:  # forward + re-integration of the accepted steps + their dgrad chains
            acc = int(ode_rl_amd.last_stats.get("n_accept", 0))
            launches = (nfe_per_step + 2 * (6 * acc + 1)) * n_convs * a.steps
        # ALGORITHMIC work of one 64->64 3x3 layer over the batch (direct-convolution FLOPs); the
        # Winograd kernel executes 2.25x fewer MFMA FLOPs for it, so `frac` is algorithmic throughput over the MFMA peak
        flop_per_launch = conv_flops(chans, a.batch) / n_convs      # average layer of f (A: every layer is 64 -> 64)
        kernel = "conv3x3_wino_kernel<4>" if a.dtype == "f32" else "conv3x3_bf16_kernel<4>"
        persistent = ode_rl_amd._lib.load().odehip_persistent_trajectory_launches() - persist0
        if persistent == a.steps and not a.train:
            # the whole trajectory is ONE launch of wino_persist_kernel: its algorithmic work = every layer of every f evaluation
            # (339.7 MFLOP per latent frame for A, T=10); duration = the timed region / steps (the copy of
            # z0, the layout kernel and two tiny fills ride along: < 2 %)
            kernel = "wino_persist_kernel"
            flop_per_launch = conv_flops(chans, a.batch) * nfe_per_step
            launches = a.steps
        per_launch_s = dev_ms * 1e-3 / launches                      # HIP events, incl. inter-kernel gaps
        achieved = flop_per_launch / per_launch_s / 1e12
        res = {
            "metric": "integrated latent frames/sec (ODEConvGRU, MovingMNIST)",
            "value": world * a.batch * T * a.steps / wall,
            "unit": "latent frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{'ODEConvGRU' if a.shape == 'A' else 'VidODE'} latents z0 (B={a.batch},{C0},16,16) per GPU, T={T} output frames "
                                   f"({T - 1} intervals), " + (f"adaptive dopri5 rtol {solver.odeint_rtol:g} atol {solver.odeint_atol:g}" if a.method == "dopri5"
                                                                else f"fixed-step {a.method} (3/8 rule)") + ", f = " + ("5x conv3x3(64->64)+ReLU" if a.shape == "A" else "conv3x3 128->64->64->128 +ReLU (VidODE)") + ", "
                                   + (("forward + adjoint backward" + (f" ({a.adjoint_norm} norm)" if a.method == "dopri5" else "") if a.adjoint else
                                       "forward + backward (discretise-then-optimise)") if a.train else "forward only (BASELINE configs[1])"),
                       "per_gpu_batch": a.batch, "frames": T, "method": a.method, "parallelism": f"batch-shard x{world}" + (", exact-global dopri5 step control" if (a.method == "dopri5" and a.global_step_control and world > 1)
                                                                     else (", per-shard dopri5 step control" if (a.method == "dopri5" and world > 1) else "")),
                       "nfe": nfe_per_step, "adjoint_stats": adj if a.method == "dopri5" else None,
                       "n_accept": int(ode_rl_amd.last_stats.get("n_accept", 0)) if a.method == "dopri5" else None},
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / (PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS),
                         "traffic": profiled_traffic() if (a.batch == 64 and a.dtype == "f32" and a.shape == "A") else None,
                         "flop_per_launch": flop_per_launch, "avg_launch_us": per_launch_s * 1e6,
                         "launches_timed": launches,
                         # `achieved` counts ALGORITHMIC (direct-convolution) FLOPs; the fp32 Winograd kernels execute 2.25x fewer
                         # on the matrix cores, so frac can exceed 1 -- the share of the MFMA peak actually executed is:
                         "executed_mfma_frac": (achieved / 2.25 / PEAK_FP32_MFMA_TFLOPS) if a.dtype == "f32" else None},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(state, z0_cpu, t_cpu, a.method, a.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
