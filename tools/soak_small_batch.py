"""Soak of round 4's small-batch paths -- both hand work between workgroups INSIDE a launch, so they are the ones that could go wrong
only now and then:
  * the SPLIT 5x5 Winograd launches (csrc/conv_wino5.hip): S workgroups per output tile, the last arriver adds the partials in split
    order.  Alternating shapes (every split count, the unsplit kernel in between: other grids, another epoch of the XCD words, the
    counters reset by the previous launch), every output compared BITWISE with the first run's;
  * the encoder loop's Euler steps on the sixteen-workgroup walk (library-owned flag area, zeroed per launch): whole training steps of
    the ODEConvGRU model at the reference's batch 4 (configs.yaml:7), loss / gradients of step k compared bitwise with a replay of step
    k from the same parameters, allocated memory constant over the last quarter of the run (before that the dopri5 workspaces follow
    the accepted-step count, which grows as the dynamics train), no persistent-launch give-up.
  python tools/soak_small_batch.py [--iters 2000] [--steps 200]"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iters", type=int, default=2000)
    p.add_argument("--steps", type=int, default=200)
    a = p.parse_args()
    import ode_rl_amd
    from ode_rl_amd import hip_ops
    dev = torch.device("cuda", 0)
    lib = ode_rl_amd._lib.load()
    # ---- split 5x5 launches
    cases = [(4, 64, 64, 128), (4, 64, 64, 64), (2, 128, 128, 256), (5, 32, 32, 64), (16, 64, 64, 64), (9, 64, 64, 128), (70, 64, 64, 128), (1, 64, 64, 128)]
    data = {}
    for b, c1, c2, co in cases:
        g = torch.Generator().manual_seed(b * 131 + co)
        w = ((torch.rand(co, c1 + c2, 5, 5, generator=g) * 2 - 1) / ((c1 + c2) * 25) ** 0.5).to(dev)
        data[(b, c1, c2, co)] = (hip_ops.nchw_to_q4(torch.randn(b, c1, 16, 16, generator=g).to(dev)), hip_ops.nchw_to_q4(torch.randn(b, c2, 16, 16, generator=g).to(dev)),
                                 hip_ops.pack_conv_weight(w), hip_ops.pack_conv_weight_winograd5(w), torch.randn(co, generator=g).to(dev), co)

    def conv(k):
        s1, s2, wp, ww, bias, co = data[k]
        return hip_ops.conv_q4(s1, wp, bias, co, 5, src2=s2, w_wino=ww)

    first = {k: conv(k).clone() for k in data}
    t0 = time.perf_counter()
    bad = 0
    for it in range(a.iters):
        k = cases[it % len(cases)]
        out = conv(k)
        if it % 16 == 0 or it > a.iters - 64:
            bad += int(not torch.equal(out, first[k]))
    torch.cuda.synchronize()
    code = lib.odehip_persistent_error(0)
    print(f"split 5x5 launches: {a.iters} launches over {len(cases)} shapes, {bad} mismatches, {time.perf_counter() - t0:.1f} s, sticky error word {code}")
    assert bad == 0 and code == 0
    # ---- training steps at batch 4 (the Euler steps of the encoder loop run on the sixteen-workgroup walk)
    import argparse as ap
    from ode_rl_amd.data import MovingMNISTSynthetic
    from ode_rl_amd.models.ODEConvGRU import ODEConvGRU
    from ode_rl_amd.optim import FusedAdam
    from ode_rl_amd.train import train_batch
    torch.manual_seed(0)
    opt = ap.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=64, in_channels=1, n_ode_layers=3, neural_ode_n_units=64,
                       neural_ode_decoder_out_ch=64, decode_diff_method="dopri5", mem=False, z_sample=False)
    m = ODEConvGRU(opt, torch.device("cpu")).to(dev)
    optim = FusedAdam(m.parameters(), lr=1e-4)
    gen = MovingMNISTSynthetic(10, 10, num_objects=[2], batch_size=4, device=dev, seed=0)
    ts = torch.arange(20, dtype=torch.float64, device=dev) / 20
    n0 = lib.odehip_persistent_trajectory_launches()
    mem0 = None
    losses, replays = [], 0
    t0 = time.perf_counter()
    for step in range(a.steps):
        batch = next(gen)
        batch.update(observed_tp=ts[:10], tp_to_predict=ts[10:])
        if step % 25 == 0:   # replay this step from a copy of the parameters: same loss and gradients, bit for bit
            snap_m, snap_o = copy.deepcopy(m.state_dict()), copy.deepcopy(optim.state_dict())
            _, _, l1, _ = train_batch(m, batch, optim)
            g1 = [q.grad.clone() for q in m.parameters()]
            m.load_state_dict(snap_m)
            optim.load_state_dict(snap_o)
            _, _, l2, _ = train_batch(m, batch, optim)
            assert torch.equal(l1, l2) and all(torch.equal(u, q.grad) for u, q in zip(g1, m.parameters())), f"step {step} is not reproducible"
            replays += 1
            losses.append(float(l2))
        else:
            losses.append(float(train_batch(m, batch, optim)[2]))
        if step == a.steps - a.steps // 4:   # the last quarter must not grow (earlier the dopri5 workspaces follow the growing step count:
            mem0 = torch.cuda.memory_allocated()   # more accepted steps -> more saved slots, up to 64)
    torch.cuda.synchronize()
    code = lib.odehip_persistent_error(0)
    print(f"training at batch 4: {a.steps} steps ({replays} replayed bitwise), loss {losses[0]:.5f} -> {losses[-1]:.5f}, "
          f"{lib.odehip_persistent_trajectory_launches() - n0} persistent launches, allocated {mem0} bytes at 3/4 of the run -> {torch.cuda.memory_allocated()} at its end, "
          f"{time.perf_counter() - t0:.1f} s, sticky error word {code}")
    assert code == 0 and all(x == x for x in losses) and losses[-1] < losses[0] and torch.cuda.memory_allocated() <= mem0 * 1.01 and torch.cuda.memory_allocated() < (4 << 30)


if __name__ == "__main__":
    main()
