"""Context numbers for BASELINE configs[3]'s model: VidODE end to end on one GPU -- forward and training
step, and the forward's parts: BatchNorm conv encoder (library), ODEConvGRUCell on 128-channel latents (HIP), DiffEqSolver (HIP),
the upsample / conv / BatchNorm flow decoder (library, one call per predicted frame: models/VidODE.py:143-158), the warp chain +
compositing (HIP).  Answers VERDICT r03 #7: which share of a VidODE step is still MIOpen.
  python tools/vidode_bench.py [--batch 64] [--frames 10] [--method rk4|dopri5] [--steps 10]"""
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
    p.add_argument("--only", default="all", choices=["all", "train"])
    p.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark: let MIOpen search solvers for the library convolutions")
    a = p.parse_args()
    torch.backends.cudnn.benchmark = a.miopen_find
    import ode_rl_amd  # noqa: F401
    from ode_rl_amd.data import MovingMNISTSynthetic
    from ode_rl_amd.models.VidODE import VidODE
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    opt = argparse.Namespace(n_downs=2, resolution=64, in_channels=1, n_layers=2, decode_diff_method=a.method)
    m = VidODE(opt, torch.device("cpu")).to(dev)
    T = a.frames
    batch = next(MovingMNISTSynthetic(T, T, num_objects=[2], batch_size=a.batch, device=dev, seed=0))
    frames = batch["observed_data"] + 0.5
    ts = torch.arange(2 * T, dtype=torch.float64, device=dev) / (2 * T)
    ones = torch.ones(a.batch, T, 1, device=dev)
    bd = {"observed_tp": ts[:T], "tp_to_predict": ts[T:], "observed_mask": ones, "mask_predicted_data": ones,
          "observed_data": frames, "data_to_predict": batch["data_to_predict"] + 0.5}
    optim = torch.optim.Adam(m.parameters(), lr=1e-4)

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
        optim.zero_grad()
        loss = m.get_loss(m.get_prediction(frames, bd), bd["data_to_predict"])
        loss.backward()
        optim.step()

    res = {"model": "VidODE", "batch": a.batch, "frames_in": T, "frames_out": T, "method": a.method}
    if a.only == "train":
        res["train_step_ms"] = timed(train, a.steps)
        print(json.dumps(res), flush=True)
        return
    res["forward_ms"] = timed(fwd, a.steps)
    res["train_step_ms"] = timed(train, a.steps)
    with torch.no_grad():
        b, t, c, h, w = frames.shape
        x = frames.reshape(b * t, c, h, w)
        res["conv_encoder_library_ms"] = timed(lambda: m.conv_encoder(x), a.steps)
        enc = m.conv_encoder(x).view(b, t, -1, 16, 16).permute(1, 0, 2, 3, 4).contiguous()
        res["odeconvgru_cell_ms"] = timed(lambda: m.encoder_z0(enc, bd["observed_tp"], None), a.steps)
        z0, _ = m.encoder_z0(enc, bd["observed_tp"], None)
        res["diffeq_solver_ms"] = timed(lambda: m.diffeq_solver(z0, bd["tp_to_predict"]), a.steps)
        sol = m.diffeq_solver(z0, bd["tp_to_predict"]).permute(1, 0, 2, 3, 4).contiguous()
        skip = m.conv_encoder(frames[:, -1])
        res["flow_decoder_library_ms"] = timed(lambda: torch.cat(m.get_flowmaps(sol_out=sol, first_prev_embed=skip, mask=None), dim=1), a.steps)
        po = torch.cat(m.get_flowmaps(sol_out=sol, first_prev_embed=skip, mask=None), dim=1)
        gx, gy = m._grids(h, w, dev)
        from ode_rl_amd.autograd import warp_composite
        res["warp_composite_ms"] = timed(lambda: warp_composite(po, frames[:, -1], gx, gy), a.steps)
    lib = res["conv_encoder_library_ms"] + res["flow_decoder_library_ms"]
    res["library_share_of_forward"] = lib / res["forward_ms"]
    # algorithmic work of the flow decoder per predicted frame and sample: up2 + conv3x3 256->128 @32x32, up2 + conv3x3 128->64 @64x64,
    # conv3x3 64->(c+3) @64x64
    flop = 2 * 9 * (256 * 128 * 1024 + 128 * 64 * 4096 + 64 * (c + 3) * 4096)
    res["flow_decoder_gflop_per_forward"] = flop * a.batch * T / 1e9
    res["flow_decoder_tflops_achieved"] = flop * a.batch * T / (res["flow_decoder_library_ms"] * 1e-3) / 1e12
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
