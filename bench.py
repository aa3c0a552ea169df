"""bench.py -- integrated latent frames/s of the Neural-ODE hot path on MI355X.

One "step" = one `DiffEqSolver.forward(z0, t)` over one batch of synthetic Moving-MNIST-shaped latents
(BASELINE.json configs[1]: ODEConvGRU latents (B,64,16,16), 10 output frames = 9 rk4(3/8) intervals, fp32).
Inputs are resident in HBM before the timed region.  With --gpus N every rank integrates its own batch
(weak scaling, no data-path collective: samples are independent under a fixed-grid solver).

`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts its own N ranks: the parent spawns
N child processes of this file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment)
BEFORE it makes any GPU call, waits for them and exits with the first non-zero exit code; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and nothing is spawned.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- dominant kernel (wino_persist_kernel: the whole trajectory, Winograd F(2x2,3x3) on exact-fp32 MFMA, in one
                  launch; conv3x3_wino_kernel<4> per layer when the persistent path is off): algorithmic FLOP per launch /
                  average launch duration measured with HIP events over the timed region;
  cpu_baseline -- the oracle (CPU restatement of torchdiffeq 0.2.1 on torch-CPU convs) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

torch = None    # imported by main() AFTER the self-launch decision: the parent of a --gpus N launch never loads torch (or any GPU runtime)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide); the bf16 conv is NOT bound by it (launch boundary, staging, epilogue)


TRAFFIC_SOURCES = ("conv_wino.hip", "persist.hip", "persist.h", "conv_common.h")   # what the headline kernel is compiled from


def source_digest():
    """sha256 over the sources of the dominant kernel: the key that ties a committed PMC summary to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, "ode-rl_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def profiled_traffic():
    """(HBM bytes per launch of the dominant kernel, provenance) from the newest committed rocprofv3 PMC summary (separate --pmc
    passes, FETCH_SIZE doubled per the gfx950 correction).  NOT measured live -- so it is only reported when the summary was
    taken on exactly the kernel sources of this tree (tools/summarize_profile.py stores their digest and the commit);
    otherwise (None, why): a stale number is dropped rather than carried along."""
    try:
        now = source_digest()
    except OSError as e:
        return None, f"kernel sources unreadable: {e!r}"
    why = "no PMC summary under profiles/"
    tags = sorted((os.path.basename(p)[:3] for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_rocprof_summary.json"))), reverse=True)
    for tag in tags:   # newest round first
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.json")) as fh:
                notes = json.load(fh)["notes"]
            val = float(notes["hbm_bytes_per_launch"])
        except Exception:
            continue
        if notes.get("source_sha256") == now:
            return val, {"profile": f"profiles/{tag}_rocprof_summary.json", "git_head": notes.get("git_head"), "source_sha256": now[:16],
                         "how": "rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE), not measured by this run"}
        why = f"profiles/{tag}_rocprof_summary.json was taken on other kernel sources (digest {str(notes.get('source_sha256'))[:16]} != {now[:16]}): dropped as stale"
        break
    return None, why


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
    p.add_argument("--max-accept", type=int, default=0,
                   help="dopri5 adjoint: accepted backward steps whose activations are kept (default: what fits 32 GiB; the mixed norm at "
                        "rtol 1e-5 needs several hundred at B=64: 280 MiB each)")
    p.add_argument("--rtol", type=float, default=None, help="dopri5 tolerances (default: DiffEqSolver's 1e-4 / 1e-5)")
    p.add_argument("--atol", type=float, default=None)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-train-leg", action="store_true", help="skip the extra forward + backward (+ gradient all-reduce) leg of the record")
    p.add_argument("--no-config0", action="store_true", help="skip the BASELINE configs[0] (B=4) GPU/CPU pair of the record")
    p.add_argument("--no-model", action="store_true", help="skip the end-to-end ODEConvGRU context object of the record")
    p.add_argument("--launch-check", action="store_true",
                   help="N ranks over gloo, no GPU work: exercises the self-launch and rendezvous path only (CPU tests)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    return p.parse_args()


def conv_flops(f_channels, batch):
    return sum(2 * batch * ci * co * 9 * 256 for ci, co in zip(f_channels[:-1], f_channels[1:]))


def host_cpu_info():
    """`lscpu` model / sockets / cores of the box the CPU baseline ran on."""
    import subprocess
    info = {}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU(s)"):
                info[k] = v
    except Exception as e:   # lscpu missing: say so rather than guess
        info["error"] = repr(e)
    return info


def cpu_baseline(state, z0, t, method, budget_s, rtol=1e-4, atol=1e-5):
    """Time the oracle on the host cores: whole trajectories of the same workload until ~budget_s.
    torch-CPU convs on 16x16 maps scale badly past a few dozen threads, so a short sweep picks the
    fastest intra-op thread count first (that count is what `cores` reports).  `value` uses the MEDIAN trajectory time."""
    from oracle import reference_modules as rm
    from oracle import torchdiffeq_ref
    ws, bs = rm.split_convnet_state(state, "gradient_net.")
    f = rm.ode_func(ws, bs)
    frames = z0.shape[0] * len(t)
    hw = torch.get_num_threads()
    best_n, best_t = hw, float("inf")
    run = lambda: torchdiffeq_ref.odeint(f, z0, t, rtol=rtol, atol=atol, method=method)   # noqa: E731
    with torch.no_grad():
        for n in sorted({hw, 64, 32, 16, 8}, reverse=True):
            if n > hw:
                continue
            torch.set_num_threads(n)
            run()  # warm-up
            t0 = time.perf_counter()
            run()
            el = time.perf_counter() - t0
            if el < best_t:
                best_n, best_t = n, el
        torch.set_num_threads(best_n)
        times, t_start = [], time.perf_counter()
        while True:
            t0 = time.perf_counter()
            run()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start >= budget_s or len(times) >= 400:
                break
    torch.set_num_threads(hw)
    times.sort()
    med = times[len(times) // 2]
    return {"value": frames / med, "unit": "latent frames/s", "cores": best_n, "kind": "port",
            "host": host_cpu_info(), "torch_threads_available": hw,
            "sample": f"{len(times)} full trajectories of the same workload (B={z0.shape[0]}, T={len(t)}, {method}), "
                      f"{sum(times):.1f} s of torch-CPU fp32, no_grad, median trajectory time, {best_n} intra-op threads (fastest of a sweep up to {hw}); "
                      "torchdiffeq is not installed: the oracle's restatement of torchdiffeq 0.2.1 is what is timed"}


def visible_gpu_count():
    """GPUs this process may use, counted WITHOUT any HIP / HSA / torch.cuda call (the launcher parent must not open the GPU
    runtime: on this pool a GPU-initialised process must not spawn-and-exec).  An explicit visibility list wins (the runtime
    would apply it to the KFD nodes anyway); otherwise the KFD topology in sysfs: a node with simd_count > 0 is a GPU.
    Returns None when neither source exists (no amdgpu driver: a CPU-only container)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            continue
    return n


def model_context(a, dev, T, method=None):
    """End-to-end ODEConvGRU (models/ODEConvGRU.py:57-88 around the hot path) at this run's batch / frames / method: predicted frames/s
    of `forward` under no_grad and of a training step (MSE loss, loss.backward(), fused Adam), Moving-MNIST-shaped frames rendered
    on the device, random-init weights of the reference's architecture (64-channel latents, 3 ODE layers).  Context, not the metric."""
    import argparse as ap
    import ode_rl_amd
    from ode_rl_amd.data import MovingMNISTSynthetic
    from ode_rl_amd.models.ODEConvGRU import ODEConvGRU
    from ode_rl_amd.optim import FusedAdam
    torch.manual_seed(0)
    method = method or a.method
    opt = ap.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=64, in_channels=1, n_ode_layers=3, neural_ode_n_units=64,
                       neural_ode_decoder_out_ch=64, decode_diff_method=method, mem=False, z_sample=False)
    m = ODEConvGRU(opt, torch.device("cpu")).to(dev)
    batch = next(MovingMNISTSynthetic(T, T, num_objects=[2], batch_size=a.batch, device=dev, seed=0))
    frames, truth = batch["observed_data"] + 0.5, batch["data_to_predict"] + 0.5      # train_test.py:180
    ts = torch.arange(2 * T, dtype=torch.float64, device=dev) / (2 * T)
    bd = {"observed_tp": ts[:T], "tp_to_predict": ts[T:]}
    optim = FusedAdam(m.parameters(), lr=1e-4)

    def timed(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def fwd():
        with torch.no_grad():
            return m(frames, bd)

    def train():
        optim.zero_grad()   # torch 2 default (set_to_none=True): what the reference's `optimizer.zero_grad()` (train_test.py:176) does today
        m.get_loss(m(frames, bd), truth).backward()
        optim.step()

    n = max(5, min(20, a.steps))
    # a training harness does not need the solver's outcome before the backward pass: the asynchronous dopri5 forward lets the host
    # enqueue decoder, loss and backward while the device still integrates (no effect on fixed-grid methods)
    was_async = ode_rl_amd.set_async_dopri5(True)
    try:
        tf, tt = timed(fwd, n), timed(train, n)
    finally:
        ode_rl_amd.set_async_dopri5(was_async)
    return {"what": "ODEConvGRU end to end (conv encoder + ODEConvGRUCell + DiffEqSolver + conv decoder), context only", "batch": a.batch,
            "frames_in": T, "frames_out": T, "method": method, "dtype": a.dtype, "steps": n,
            "forward": {"ms": tf * 1e3, "value": a.batch * T / tf, "unit": "predicted frames/s"},
            "train_step": {"ms": tt * 1e3, "value": a.batch * T / tt, "unit": "predicted frames/s",
                           "what": "MSE loss + loss.backward() + fused Adam step (ode_rl_amd.optim.FusedAdam)"}}


def self_launch(a, grace_s=10.0):
    """--gpus N without a launcher: one child process of this file per rank.  The parent makes NO GPU-runtime call -- torch is
    not even imported here (`tests/test_bench_launch.py` asserts it) -- and never replaces itself: children are spawned, waited
    for, and the first non-zero exit code is propagated.  Rank 0's stdout (the JSON line) is inherited.  A rank that fails makes
    the parent terminate the others (they would wait at the rendezvous for ever) and, if one ignores SIGTERM for `grace_s`
    seconds (stuck in a GPU wait), kill it: the parent always exits."""
    import socket
    import subprocess
    assert "torch" not in sys.modules or os.environ.get("ODEHIP_BENCH_ALLOW_TORCH_IN_PARENT"), "the launcher parent must not load torch"
    n = a.gpus
    if not a.launch_check and not os.environ.get("ODEHIP_BENCH_REHEARSAL"):
        have = visible_gpu_count()
        if have is not None and have < n:
            raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible (set ODEHIP_BENCH_REHEARSAL=1 to rehearse {n} ranks on one GPU over gloo)")
        # have is None: no sysfs topology to count from -- every rank checks LOCAL_RANK < device_count() itself and exits non-zero
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL's intra-node transport exchanges buffers between the ranks' processes through HIP IPC
        # handles, and this pool's host driver only supports the dmabuf flavour (legacy mode fails with `hipIpcGetMemHandle: invalid
        # argument`).  The image exports it already; setdefault keeps a caller's own value and only fills the gap
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    deadline = None
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # a failed rank would leave the others waiting at the rendezvous: stop exactly those PIDs
                    q.terminate()
                deadline = time.monotonic() + grace_s
        if deadline is not None and alive and time.monotonic() > deadline:
            for q in alive:              # SIGTERM ignored (a rank stuck in a GPU wait): SIGKILL the exact PIDs we started
                q.kill()
            deadline = time.monotonic() + grace_s
    raise SystemExit(rc)


def collective_identity(dist_mod, rank, local, world, dev, rehearsal):
    """What the process group really is, gathered from every rank: backend, the group's own world size, and one device identity
    per rank (index, name, PCI domain:bus:device, uuid) -- so a scaling line proves N distinct GPUs took part."""
    pr = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local, "device_index": dev.index, "name": pr.name, "arch": getattr(pr, "gcnArchName", None),
          "pci": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}",
          "uuid": str(getattr(pr, "uuid", "")), "pid": os.getpid()}
    if dist_mod is None:
        return {"backend": None, "world_size": 1, "ranks": [me], "distinct_devices": 1}
    allr = [None] * world
    dist_mod.all_gather_object(allr, me)
    ids = {(r["pci"], r["uuid"]) for r in allr}
    if not rehearsal and len(ids) != world:
        raise SystemExit(f"--gpus {world}: the ranks do not sit on {world} distinct GPUs: {allr}")
    return {"backend": dist_mod.get_backend(), "world_size": dist_mod.get_world_size(), "ranks": allr, "distinct_devices": len(ids),
            "rehearsal_one_gpu": bool(rehearsal)}


def launch_check(rank, world):
    """Rendezvous + one barrier + one MAX all-reduce over gloo; rank 0 prints a JSON line.  No GPU work."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    x = torch.tensor([float(rank)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    ids = [None] * world
    dist.all_gather_object(ids, {"rank": rank, "pid": os.getpid()})
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "max_rank_seen": int(x.item()),
                          "collective": {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                         "ranks": [i["rank"] for i in ids], "distinct_pids": len({i["pid"] for i in ids})}}), flush=True)
    dist.destroy_process_group()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)     # does not return
    global torch
    import torch           # only a rank (or the single-GPU run) loads torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.launch_check:
        return launch_check(rank, world)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("ODEHIP_BENCH_REHEARSAL"):   # N ranks on ONE GPU over gloo: exercises this file's N > 1 path on a 1-GPU box
            local = 0
            # the persistent trajectory kernel holds every CU and assumes its process owns the GPU (one process per GPU): two of
            # them sharing a card could each hold half the CUs waiting for the other half
            os.environ["ODEHIP_PERSISTENT"] = "0"
            dist.init_process_group("gloo")
        else:
            have = torch.cuda.device_count()
            if local >= have:
                raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {have} GPU(s) visible")
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rehearsal = bool(os.environ.get("ODEHIP_BENCH_REHEARSAL")) and world > 1
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
        f.zero_grad()   # torch 2 default (set_to_none=True), what the reference's `optimizer.zero_grad()` (train_test.py:176) does today
        if a.adjoint:
            o = ode_rl_amd.odeint_adjoint(f, zz, t, rtol=solver.odeint_rtol, atol=solver.odeint_atol, method=a.method,
                                          adjoint_options=({"norm": a.adjoint_norm, **({"max_accept": a.max_accept} if a.max_accept else {})}
                                                           if a.method == "dopri5" else None))
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
    def measure(step_fn, steps, warmup):
        """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; wall = MAX over ranks.
        HIP events (on the stream the kernels are launched on: torch's current stream, which hip_ops hands to the C ABI) bracket
        the region.  The MEDIAN step time comes from a second, untimed-for-the-record pass with one event between steps: an event
        record inside the timed region costs a few per cent (same-box A/B), so the region the record is computed from has none."""
        for _ in range(warmup):
            o = step_fn()
        sync()
        p0 = lib.odehip_persistent_trajectory_launches()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            o = step_fn()
        ev1.record()
        sync()
        wall = time.perf_counter() - t0
        n_persist = lib.odehip_persistent_trajectory_launches() - p0
        wt = torch.tensor([wall], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        m = min(steps, 20)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(m + 1)]
        evs[0].record()
        for i in range(m):
            step_fn()
            evs[i + 1].record()
        sync()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(m))
        return o, float(wt.item()), ev0.elapsed_time(ev1), per[len(per) // 2], n_persist

    lib = ode_rl_amd._lib.load()
    collective = collective_identity(dist, rank, local, world, dev, rehearsal)
    out, wall, dev_ms, median_ms, persistent = measure(step, a.steps, a.warmup)
    assert out.shape == (T, a.batch, C0, 16, 16) and bool(torch.isfinite(out).all())
    fwd_stats = dict(ode_rl_amd.last_stats) if a.method == "dopri5" else {}
    adj = dict(ode_rl_amd.last_adjoint_stats) if (a.method == "dopri5" and a.train and a.adjoint) else None

    # ---- second leg of the record: the training step of the same workload (forward + backward through the solver + the ONE
    # gradient all-reduce of north_star when N > 1), so the collective sits inside a timed region of every scaling run
    train_leg = None
    if not a.train and not a.no_train_leg and not a.graph:
        gout2 = torch.randn(T, a.batch, C0, 16, 16, generator=torch.Generator().manual_seed(99)).to(dev)
        from ode_rl_amd.dist import allreduce_gradients

        def train_step():
            zz = z0.detach().requires_grad_(True)
            f.zero_grad()   # torch 2 default (set_to_none=True), what the reference's `optimizer.zero_grad()` (train_test.py:176) does today
            o = solver(zz, t)
            o.backward(gout2)
            if dist is not None:
                allreduce_gradients(f.parameters())
            return o.detach()
        ksteps = max(5, a.steps // 2)
        _, twall, _, tmed, _ = measure(train_step, ksteps, 2)
        train_leg = {"what": "forward + backward through the solver (discretise-then-optimise)" + (" + one flattened RCCL all-reduce of the gradients" if dist is not None else ""),
                     "value": world * a.batch * T * ksteps / twall, "unit": "latent frames/s", "steps": ksteps,
                     "ms_per_step": twall / ksteps * 1e3, "median_ms_per_step": tmed,
                     "allreduce_bytes": 4 * sum(p.numel() for p in f.parameters()) if dist is not None else 0}

    # ---- BASELINE configs[0] (the reference's own CPU-runnable case: B=4, same grid and method) on this GPU, next to its CPU time
    config0 = None
    if rank == 0 and world == 1 and not a.no_config0 and not a.train and a.batch != 4 and a.shape == "A" and not a.graph:
        z4_cpu = torch.randn(4, C0, 16, 16, generator=torch.Generator().manual_seed(1234)) * 0.5
        z4 = z4_cpu.to(dev)

        def step4():
            with torch.no_grad():
                return solver(z4, t)
        _, w4, _, m4, _ = measure(step4, 20, 5)
        config0 = {"workload": f"BASELINE configs[0]: B=4, T={T}, {a.method}, {a.dtype}", "gpu": {"value": 4 * T * 20 / w4, "unit": "latent frames/s", "ms_per_step": w4 / 20 * 1e3, "median_ms_per_step": m4}}
        # (its CPU side is timed at the very end, with the other CPU leg: the oracle's intra-op thread pool -- up to 128 threads on this
        # host -- keeps spinning for a while after a run, and the host-driven parts of the GPU legs that follow, the dopri5 model step
        # most of all, then measured 14.7 instead of 9.5 ms)

    # ---- context ("also report end-to-end ODEConvGRU.forward frames/s"): the WHOLE model of models/ODEConvGRU.py around the
    # path -- conv encoder, ODEConvGRUCell, DiffEqSolver, conv decoder -- forward, and one training step (MSE loss, backward, Adam)
    model_ctx = None
    if rank == 0 and world == 1 and not a.no_model and not a.train and a.shape == "A" and not a.graph:
        try:
            model_ctx = model_context(a, dev, T)
            if a.method != "dopri5":   # ... and with the reference's DEFAULT decoder solver (configs.yaml:79 decode_diff_method 'dopri5')
                model_ctx["with_reference_default_solver_dopri5"] = {k: v for k, v in model_context(a, dev, T, "dopri5").items()
                                                                     if k in ("method", "forward", "train_step")}
        except Exception as e:   # context only (library convolutions either side of the path): never fails the record
            model_ctx = {"error": repr(e)[:300]}

    if rank == 0:
        if a.method == "dopri5":
            nfe_per_step = int(fwd_stats.get("nfe", 0))
        else:
            nfe_per_step = {"rk4": 4, "midpoint": 2, "euler": 1}[a.method] * (T - 1)
        n_convs = len(chans) - 1
        F_f = conv_flops(chans, a.batch)                  # ALGORITHMIC FLOPs of one evaluation of f over the batch
        fused_bf16 = a.dtype == "bf16" and a.shape == "A"    # 64 -> 64 stacks: fstack_bf16_kernel runs a whole f per launch
        note = None
        if a.train:
            # a training step runs forward, input-gradient and weight-gradient kernels: no single kernel dominates, so the
            # roofline entry is the whole step -- algorithmic work (forward F, dgrad F, wgrad F per evaluation) over its duration
            if a.method == "dopri5" and a.adjoint:
                evals = nfe_per_step + 3 * adj.get("nfe", 0)                      # augmented evaluation = forward + dgrad + wgrad
            elif a.method == "dopri5":
                evals = nfe_per_step + 3 * (6 * int(fwd_stats.get("n_accept", 0)) + 1)   # forward, then re-integration + reverse sweep
            elif a.adjoint:
                evals = nfe_per_step + 3 * nfe_per_step
            else:
                evals = 3 * nfe_per_step
            kernel, launches, flop_per_launch = "whole training step (forward + input-gradient + weight-gradient kernels)", a.steps, F_f * evals
            note = "launch = one training step; achieved = algorithmic FLOPs of all its conv work / step duration"
        elif persistent == a.steps and a.dtype == "f32":
            # the whole trajectory is ONE launch of wino_persist_kernel: its algorithmic work = every layer of every f evaluation
            # (339.7 MFLOP per latent frame for A, T=10); duration = the timed region / steps (the copy of
            # z0, the layout kernel and two tiny fills ride along: < 2 %)
            # (batches up to 16 walk with sixteen workgroups per sample: wino_persist16_kernel)
            small16 = a.shape == "A" and a.batch <= 16 and os.environ.get("ODEHIP_PERSIST16") != "0"
            kernel, launches, flop_per_launch = ("wino_persist16_kernel" if small16 else ("wino_persist_kernel" if a.shape == "A" else "wino_persist_v_kernel")), a.steps, F_f * nfe_per_step
        elif fused_bf16 and persistent == a.steps and a.method != "dopri5":
            # bf16, 64-channel stack: the whole trajectory is ONE launch of ftraj_bf16_kernel (one workgroup per sample)
            kernel, launches, flop_per_launch = "ftraj_bf16_kernel", a.steps, F_f * nfe_per_step
        elif a.method == "dopri5" and persistent > 0 and a.dtype == "f32":
            # dopri5 forward: every attempted step (6 evaluations of f) is one launch of wino_persist_kernel; the two evaluations
            # of the initial-step search run as per-layer launches and are left out of this entry
            small16 = a.shape == "A" and a.batch <= 16 and os.environ.get("ODEHIP_PERSIST16") != "0"
            kernel, launches, flop_per_launch = ("wino_persist16_kernel" if small16 else ("wino_persist_d_kernel" if a.shape == "A" else "wino_persist_v_kernel")) + " (one launch per attempted step)", persistent, 6 * F_f
            note = "dev time of the whole region / persistent launches; the 2 evaluations of the initial-step search ride along"
        elif fused_bf16:
            kernel, launches, flop_per_launch = "fstack_bf16_kernel", nfe_per_step * a.steps, F_f      # one launch per evaluation of f
        else:
            kernel = "conv3x3_wino_kernel<4>" if a.dtype == "f32" else "conv3x3_bf16_kernel"
            launches, flop_per_launch = nfe_per_step * n_convs * a.steps, F_f / n_convs             # one launch per layer
        per_launch_s = dev_ms * 1e-3 / launches                      # HIP events over the timed region, incl. inter-kernel gaps
        achieved = flop_per_launch / per_launch_s / 1e12
        peak = PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        traffic, traffic_src = (profiled_traffic() if (a.batch == 64 and a.dtype == "f32" and a.shape == "A" and kernel == "wino_persist_kernel")
                                else (None, "PMC passes exist for the headline workload only"))
        res = {
            "metric": "integrated latent frames/sec (ODEConvGRU, MovingMNIST)",
            "value": world * a.batch * T * a.steps / wall,
            "unit": "latent frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3, "median_ms_per_step": median_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{'ODEConvGRU' if a.shape == 'A' else 'VidODE'} latents z0 (B={a.batch},{C0},16,16) per GPU, T={T} output frames "
                                   f"({T - 1} intervals), " + (f"adaptive dopri5 rtol {solver.odeint_rtol:g} atol {solver.odeint_atol:g}" if a.method == "dopri5"
                                                                else f"fixed-step {a.method}" + (" (3/8 rule)" if a.method == "rk4" else "")) + ", f = " + ("5x conv3x3(64->64)+ReLU" if a.shape == "A" else "conv3x3 128->64->64->128 +ReLU (VidODE)") + ", "
                                   + (("forward + adjoint backward" + (f" ({a.adjoint_norm} norm)" if a.method == "dopri5" else "") if a.adjoint else
                                       "forward + backward (discretise-then-optimise)") if a.train else "forward only" + (" (BASELINE configs[1])" if (a.batch, T, a.method, a.dtype, a.shape) == (64, 10, "rk4", "f32", "A") else "")),
                       "per_gpu_batch": a.batch, "frames": T, "method": a.method, "parallelism": f"batch-shard x{world}" + (", exact-global dopri5 step control" if (a.method == "dopri5" and a.global_step_control and world > 1)
                                                                     else (", per-shard dopri5 step control" if (a.method == "dopri5" and world > 1) else "")),
                       "nfe": nfe_per_step, "adjoint_stats": adj if a.method == "dopri5" else None,
                       "n_accept": int(fwd_stats.get("n_accept", 0)) if a.method == "dopri5" else None},
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_provenance": traffic_src,   # (PMC passes exist for the headline only)
                         "flop_per_launch": flop_per_launch, "avg_launch_us": per_launch_s * 1e6,
                         "launches_timed": launches, "note": note,
                         # `achieved` counts ALGORITHMIC (direct-convolution) FLOPs; the fp32 Winograd kernels execute 2.25x fewer
                         # on the matrix cores, so frac can exceed 1 -- the share of the MFMA peak actually executed is:
                         "executed_mfma_frac": (achieved / 2.25 / PEAK_FP32_MFMA_TFLOPS) if (a.dtype == "f32" and not a.train) else None},
            "train": train_leg, "config0": config0, "model": model_ctx, "collective": collective,
        }
        if world == 1 and not a.no_cpu_baseline and config0 is not None:
            config0["cpu"] = cpu_baseline(state, z4_cpu, t_cpu, a.method, 3.0, rtol=solver.odeint_rtol, atol=solver.odeint_atol)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(state, z0_cpu, t_cpu, a.method, a.cpu_seconds, rtol=solver.odeint_rtol, atol=solver.odeint_atol)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
