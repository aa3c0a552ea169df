"""Soak of the frame codec's backward kernels (csrc/frame_codec_backward.hip): the same inputs N times, alternating with other
batch shapes in between (other grids, other workspace addresses), every gradient compared BITWISE with the first run's, plus the
saved-intermediate and the recomputing variant of the decoder against each other.
  python tools/soak_codec.py [--iters 300]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iters", type=int, default=300)
    a = p.parse_args()
    from conftest import procedural_state_dict, procedural_tensor
    from ode_rl_amd import hip_ops
    from ode_rl_amd.models.ODEConvGRU import Decoder, Encoder
    dev = torch.device("cuda", 0)
    shapes = [(64, 64, 10), (64, 4, 10), (32, 23, 7), (64, 1, 1)]   # latent channels, B, T
    mods = {}
    for lat in (32, 64):
        e, d = Encoder(1, lat, 2, nonlinear="leaky_relu"), Decoder(lat, 1, 2, nonlinear="leaky_relu")
        e.load_state_dict(procedural_state_dict(e.state_dict(), 91))
        d.load_state_dict(procedural_state_dict(d.state_dict(), 92))
        mods[lat] = (e.to(dev), d.to(dev))
    data = {}
    for lat, b, t in shapes:
        data[(lat, b, t)] = tuple(x.to(dev) for x in (procedural_tensor((b, t, 1, 64, 64), 191, -1, 1), procedural_tensor((t, b, lat, 16, 16), 192, -2, 2),
                                                        procedural_tensor((t, b, lat, 16, 16), 193, -1, 1), procedural_tensor((t, b, 1, 64, 64), 194, -1, 1)))

    def run(key):
        lat = key[0]
        enc, dec = mods[lat]
        frames, z, g_enc, g_pred = data[key]
        enc.zero_grad(set_to_none=True)
        dec.zero_grad(set_to_none=True)
        hip_ops.frame_encode_autograd(enc.encoder, frames).backward(g_enc)
        zc = z.clone().requires_grad_(True)
        hip_ops.frame_decode_autograd(dec.decoder, zc, True).backward(g_pred)
        return [q.grad.clone() for q in enc.parameters()] + [q.grad.clone() for q in dec.parameters()] + [zc.grad.clone()]

    first = {k: run(k) for k in data}
    os.environ["ODEHIP_CODEC_SAVE_MID"] = "0"
    for k in data:
        assert all(torch.equal(u, v) for u, v in zip(first[k], run(k))), f"saved vs recomputed intermediate differ at {k}"
    os.environ["ODEHIP_CODEC_SAVE_MID"] = "1"
    t0, bad, launches = time.time(), 0, 0
    keys = list(data)
    for it in range(a.iters):
        k = keys[it % len(keys)]
        got = run(k)
        launches += 7
        if not all(torch.equal(u, v) for u, v in zip(first[k], got)):
            bad += 1
            print(f"iteration {it}: shape {k} differs from its first run", flush=True)
        if it % 100 == 99:
            torch.cuda.synchronize()
            print(f"{it + 1} iterations, {time.time() - t0:.0f} s, {bad} mismatches", flush=True)
    torch.cuda.synchronize()
    print(f"soak_codec: {a.iters} backward passes over {len(keys)} shapes ({launches} kernel launches of the backward), {bad} mismatches, "
          f"saved == recomputed intermediate: yes")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
