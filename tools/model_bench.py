"""Context numbers: end-to-end ODEConvGRU forward and training step on one GPU, with the time of each
part (conv encoder / ODEConvGRUCell / DiffEqSolver / conv decoder).  Synthetic Moving-MNIST frames rendered on the device (ode-rl_amd/data.py).
  python tools/model_bench.py [--batch 64] [--frames 10] [--method rk4] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--frames", type=int, default=10)
    p.add_argument("--method", default="rk4")
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="compute dtype of the hot-path convs")
    p.add_argument("--only", default="all", choices=["all", "train"], help="'train': time the training step alone (profiling)")
    p.add_argument("--sync-solver", action="store_true", help="dopri5: the synchronous forward (the host waits for the controller inside the model's forward)")
    p.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark: let MIOpen search solvers for the convs either side of the path")
    a = p.parse_args()
    torch.backends.cudnn.benchmark = a.miopen_find
    import ode_rl_amd
    from ode_rl_amd.models.ODEConvGRU import ODEConvGRU
    if a.dtype == "bf16":
        ode_rl_amd.set_compute_dtype("bf16")
    ode_rl_amd.set_async_dopri5(not a.sync_solver)   # a training harness: the solver's outcome is only needed at the backward pass
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    opt = argparse.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=64, in_channels=1, n_ode_layers=3,
                             neural_ode_n_units=64, neural_ode_decoder_out_ch=64, decode_diff_method=a.method, mem=False,
                             z_sample=False)
    m = ODEConvGRU(opt, torch.device("cpu")).to(dev)
    T = a.frames
    from ode_rl_amd.data import MovingMNISTSynthetic
    batch = next(MovingMNISTSynthetic(T, T, num_objects=[2], batch_size=a.batch, device=dev, seed=0))  # rendered on the device
    frames, truth = batch["observed_data"] + 0.5, batch["data_to_predict"] + 0.5                     # train_test.py:180
    ts = torch.arange(2 * T, dtype=torch.float64, device=dev) / (2 * T)
    bd = {"observed_tp": ts[:T], "tp_to_predict": ts[T:]}
    from ode_rl_amd.optim import FusedAdam
    optim = FusedAdam(m.parameters(), lr=1e-4)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def fwd():
        with torch.no_grad():
            return m(frames, bd)

    def train():
        optim.zero_grad()   # torch 2 default (set_to_none=True): what the reference's `optimizer.zero_grad()` (train_test.py:176) does today
        loss = m.get_loss(m(frames, bd), truth)
        loss.backward()
        optim.step()

    res = {"batch": a.batch, "frames_in": T, "frames_out": T, "method": a.method, "dtype": a.dtype, "async_dopri5": not a.sync_solver}
    if a.only == "train":
        res["train_step_ms"] = timed(train, a.steps)
        print(json.dumps(res), flush=True)
        return
    res["forward_ms"] = timed(fwd, a.steps)
    res["train_step_ms"] = timed(train, a.steps)
    # parts of the forward
    with torch.no_grad():
        b, t, c, h, w = frames.shape
        x = frames.view(b * t, c, h, w)
        res["conv_encoder_ms"] = timed(lambda: m.conv_encoder.encode_time_first(frames), a.steps)       # fused launch
        res["conv_encoder_library_ms"] = timed(lambda: m.conv_encoder(x), a.steps)                     # torch -> MIOpen
        enc = m.conv_encoder.encode_time_first(frames)
        res["odeconvgru_cell_ms"] = timed(lambda: m.ode_convgru_cell(enc, bd["observed_tp"]), a.steps)
        z0, _ = m.ode_convgru_cell(enc, bd["observed_tp"])
        res["diffeq_solver_ms"] = timed(lambda: m.diffeq_solver(z0, bd["tp_to_predict"]), a.steps)
        sol = m.diffeq_solver(z0, bd["tp_to_predict"])
        s2 = sol.view(-1, *sol.shape[2:])
        res["conv_decoder_ms"] = timed(lambda: m.conv_decoder.decode_sigmoid(sol), a.steps)             # fused launch
        res["conv_decoder_library_ms"] = timed(lambda: torch.sigmoid(m.conv_decoder(s2)), a.steps)     # torch -> MIOpen
    res["pred_frames_per_s_forward"] = a.batch * T / (res["forward_ms"] * 1e-3)
    res["pred_frames_per_s_train"] = a.batch * T / (res["train_step_ms"] * 1e-3)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
