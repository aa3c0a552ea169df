"""Moving-MNIST-shaped frames rendered on the device (`odehip_mmnist_render`, ode-rl_amd/data.py) against
the CPU restatement of the reference's generator (oracle/moving_mnist_ref.py <- dataloader.py:47-103, :217-218).
Integer/byte work: the GPU frames must be BIT-EXACT."""
import zlib

import numpy as np
import pytest
import torch

from oracle import moving_mnist_ref as mm


def test_oracle_trajectory_hand_checked():
    # heading +x from x = 0.95: 1.05 -> clamped to 1.0 and reflected; then 0.9, 0.8, 0.7 (canvas 36, truncation)
    sy, sx = mm.trajectory(0.95, 0.5, 0.0, 4)
    assert sx.tolist() == [36, 32, 28, 25] and sy.tolist() == [18, 18, 18, 18]
    # heading -x from x = 0.05: -0.05 -> clamped to 0 and reflected
    sy, sx = mm.trajectory(0.05, 0.5, np.pi, 4)
    assert sx.tolist() == [0, 3, 7, 10]
    rng = np.random.default_rng(1)
    for _ in range(50):
        sy, sx = mm.trajectory(rng.random(), rng.random(), rng.random() * 2 * np.pi, 60)
        assert sy.min() >= 0 and sx.min() >= 0 and sy.max() <= 36 and sx.max() <= 36  # the digit never leaves the canvas


def test_oracle_render_compositing_and_range():
    glyphs = np.zeros((2, 28, 28), np.uint8)
    glyphs[0, :, :] = 100
    glyphs[1, 10:20, 10:20] = 255
    obs, pred = mm.render(glyphs, [0, 1], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0], 2, 1)
    assert obs.shape == (2, 1, 64, 64) and pred.shape == (1, 1, 64, 64) and obs.dtype == np.float32
    f = obs[0, 0]  # after one step both digits sit at left = int(36 * 0.1) = 3, top = 0
    assert f[0, 3] == np.float32(100) / np.float32(255) - np.float32(0.5) and f[15, 3 + 15] == np.float32(0.5) and f[40, 40] == np.float32(-0.5)
    assert f.min() == -0.5 and f.max() == 0.5


def test_synthetic_glyphs_are_deterministic_and_distinct():
    from ode_rl_amd import data
    g = data.synthetic_digit_glyphs()
    assert g.shape == (10, 28, 28) and g.dtype == np.uint8
    assert zlib.crc32(g.tobytes()) == 2594711932
    flat = g.reshape(10, -1).astype(np.int32)
    for i in range(10):
        for j in range(i + 1, 10):
            assert np.abs(flat[i] - flat[j]).sum() > 1000


def test_get_next_batch_times():
    from ode_rl_amd import data
    d = {"observed_data": torch.zeros(2, 10, 1, 64, 64), "data_to_predict": torch.zeros(2, 20, 1, 64, 64)}
    b = data.get_next_batch(d)
    assert b["timesteps"].dtype == torch.float64 and len(b["observed_tp"]) == 10 and len(b["tp_to_predict"]) == 20
    assert torch.equal(b["timesteps"], torch.arange(30, dtype=torch.float64) / 30)


def test_generator_needs_a_gpu():
    from ode_rl_amd import data
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        data.MovingMNISTSynthetic(10, 10, device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("n_digits,t_in,t_out", [(1, 10, 10), (2, 10, 10), (3, 20, 40), (2, 0, 5)])
def test_device_frames_bit_exact(cuda, n_digits, t_in, t_out):
    from ode_rl_amd import data
    gen = data.MovingMNISTSynthetic(t_in, t_out, num_objects=[n_digits], batch_size=5, device=cuda, seed=n_digits)
    st = gen.draw()
    # edge cases the walk must reproduce: starts on the walls / in the corners, axis-aligned headings
    st["x"][0, 0], st["y"][0, 0], st["theta"][0, 0] = 0.0, 0.0, np.pi * 1.25
    st["x"][1, 0], st["y"][1, 0], st["theta"][1, 0] = 0.999, 0.999, 0.25 * np.pi
    st["x"][2, 0], st["y"][2, 0], st["theta"][2, 0] = 0.5, 0.5, 0.5 * np.pi
    obs, pred = gen.render(st)
    torch.cuda.synchronize()
    for b in range(5):
        ro, rp = mm.render(gen.glyphs_host, st["ids"][b], st["x"][b], st["y"][b], st["theta"][b], t_in, t_out)
        assert np.array_equal(obs[b].cpu().numpy(), ro), f"observed frames differ for sample {b}"
        assert np.array_equal(pred[b].cpu().numpy(), rp), f"frames to predict differ for sample {b}"


@pytest.mark.gpu
def test_device_frames_match_the_reference_generator(cuda):
    """tests/golden/mmnist.npz = frames the REFERENCE's MovingMNIST produced under a seeded `random` (dataloader.py:47-103,
    :188-223) on the build's glyphs, with the draws it consumed: the device renderer must reproduce them bit for bit."""
    from conftest import load_golden
    from ode_rl_amd import data
    g = load_golden("mmnist.npz")
    for case, (n_in, n_out) in enumerate(((10, 10), (20, 40), (3, 2))):
        draws = g[f"case{case}.draws"]                      # (sample, digit, [x, y, theta, glyph id])
        gen = data.MovingMNISTSynthetic(n_in, n_out, num_objects=[draws.shape[1]], batch_size=draws.shape[0], device=cuda)
        st = {"x": draws[:, :, 0].copy(), "y": draws[:, :, 1].copy(), "theta": draws[:, :, 2].copy(), "ids": draws[:, :, 3].astype(np.int32)}
        obs, pred = gen.render(st)
        torch.cuda.synchronize()
        assert np.array_equal(obs.cpu().numpy(), g[f"case{case}.observed"]), f"case {case}: observed frames differ"
        assert np.array_equal(pred.cpu().numpy(), g[f"case{case}.to_predict"]), f"case {case}: frames to predict differ"


@pytest.mark.gpu
def test_device_frames_full_size_properties(cuda):
    from ode_rl_amd import data
    gen = data.MovingMNISTSynthetic(20, 40, num_objects=[2], batch_size=64, device=cuda, seed=7)
    batch = next(gen)
    obs, pred = batch["observed_data"], batch["data_to_predict"]
    assert obs.shape == (64, 20, 1, 64, 64) and pred.shape == (64, 40, 1, 64, 64)
    frames = torch.cat([obs, pred], 1)
    assert float(frames.min()) == -0.5 and float(frames.max()) <= 0.5
    lit = (frames > -0.5).flatten(2).sum(-1)          # lit pixels per frame: at most two glyphs, at least the larger one's overlap
    per_glyph = torch.tensor([(g > 0).sum() for g in gen.glyphs_host])
    assert int(lit.max()) <= 2 * int(per_glyph.max()) and int(lit.min()) >= int(per_glyph.min())
    assert not torch.equal(frames[:, 0], frames[:, 1])  # the digits move
    nxt = next(gen)
    assert not torch.equal(nxt["observed_data"], obs) and int(nxt["idx"][0]) == 64


@pytest.mark.gpu
def test_train_batch_on_generated_frames(cuda):
    import argparse
    import ode_rl_amd  # noqa: F401
    from ode_rl_amd import data
    from ode_rl_amd.models.ODEConvGRU import ODEConvGRU
    from ode_rl_amd.optim import FusedAdam
    from ode_rl_amd.train import train_batch
    torch.manual_seed(0)
    opt = argparse.Namespace(resolution=64, n_downs=2, conv_encoder_out_ch=64, in_channels=1, n_ode_layers=3, neural_ode_n_units=64,
                             neural_ode_decoder_out_ch=64, decode_diff_method="rk4", mem=False, z_sample=False)
    model = ODEConvGRU(opt, torch.device("cpu")).to(cuda)
    optim = FusedAdam(model.parameters(), lr=1e-3)
    gen = data.MovingMNISTSynthetic(5, 5, batch_size=4, device=cuda, seed=3)
    losses = []
    for _ in range(6):
        bd = data.get_next_batch(next(gen))
        _, _, loss, _ = train_batch(model, bd, optim)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_train_batch_repeats_a_sealed_asynchronous_step_from_clean_buffers():
    """train_batch(async_solver=True): a forward whose dopri5 solve was sealed (AsyncSolveTruncated at the backward pass) has fed NaN frames
    to everything behind the solver -- a BatchNorm there folds them into its running statistics.  The repeat must start from the
    buffers of before the sealed pass and take exactly one optimiser step.  (Host logic only: a stub model on the CPU.)"""
    from ode_rl_amd import _lib, hip_ops
    from ode_rl_amd.train import train_batch

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))
            self.bn = torch.nn.BatchNorm1d(3)
            self.calls, self.async_seen = 0, []

        def get_prediction(self, inp, batch_dict=None):
            self.calls += 1
            self.async_seen.append(hip_ops._async_dopri5)
            x = inp * self.w
            if self.calls == 1:   # the sealed pass: NaN frames reach the BatchNorm, the backward pass reports the truncation
                self.bn(x * float("nan"))

                class Sealed(torch.autograd.Function):
                    @staticmethod
                    def forward(ctx, v):
                        return v.clone()

                    @staticmethod
                    def backward(ctx, g):
                        raise _lib.AsyncSolveTruncated("sealed")
                return Sealed.apply(x)
            return self.bn(x)

        def get_loss(self, pred, truth):
            return ((pred - truth) ** 2).mean()

    m = Stub()
    optim = torch.optim.SGD(m.parameters(), lr=0.1)
    bd = {"observed_data": torch.randn(8, 3), "data_to_predict": torch.randn(8, 3)}
    was = hip_ops._async_dopri5
    _, _, loss, _ = train_batch(m, bd, optim, async_solver=True)
    assert m.calls == 2 and m.async_seen == [True, was]          # the repeat runs on the caller's (synchronous) setting
    assert hip_ops._async_dopri5 == was
    assert torch.isfinite(loss) and torch.isfinite(m.bn.running_mean).all() and torch.isfinite(m.bn.running_var).all()
    assert int(m.bn.num_batches_tracked) == 1                    # one step's worth of statistics, not two
    assert torch.isfinite(m.w).all() and not torch.equal(m.w.detach(), torch.ones(3))


def test_async_dopri5_attempt_budget_follows_the_previous_solve(monkeypatch):
    """The policy of ode_rl_amd.hip_ops.PendingDopri5.collect (host logic, the library call stubbed): the next asynchronous solve
    enqueues what the last one used plus a quarter (at least 2 more, at least 4, at most ASYNC_ATTEMPTS_MAX), keeps its budget while
    the need stays within [half, all] of it, doubles on a sealed solve (AsyncSolveTruncated), and a failed solve raises the same error
    at every later look."""
    from ode_rl_amd import _lib, hip_ops
    real = _lib.load()
    script = []   # (rc, n_accept, n_reject, enqueued) per collect call

    class Stub:
        def __getattr__(self, name):
            return getattr(real, name)

        def odehip_odeint_dopri5_collect(self, token, stats, log, cap, saved):
            rc, acc, rej, enq = script.pop(0)
            stats[0], stats[1], stats[2], stats[3] = 2 + 6 * (acc + rej), acc, rej, enq
            return rc

        def odehip_last_error(self):
            return b"stub: sealed"

    monkeypatch.setattr(_lib, "load", lambda: Stub())
    monkeypatch.setattr(hip_ops, "_async_attempts", 8)
    monkeypatch.setattr(hip_ops, "_pending_solves", [])

    def collect(rc, acc, rej):
        script.append((rc, acc, rej, hip_ops._async_attempts))
        p = hip_ops.PendingDopri5(0, None, 0, None)
        return p, p.collect

    p, c = collect(0, 2, 0)            # used 2 of 8: want 4, exactly half of the budget -> the budget stays
    st, saved = c()
    assert st["n_accept"] == 2 and saved is None and hip_ops._async_attempts == 8 and not hip_ops._pending_solves
    assert c() is p._result            # a second look does not call the library again (the script is empty)
    collect(0, 1, 0)[1]()              # used 1: want 3, less than half of 8 -> follow it down, never below 4
    assert hip_ops._async_attempts == 4
    collect(0, 3, 0)[1]()              # used 3 of 4: within [half, all] -> want 5 > 4 -> 5
    assert hip_ops._async_attempts == 5
    collect(0, 3, 1)[1]()              # used 4: want 6
    assert hip_ops._async_attempts == 6
    collect(0, 3, 0)[1]()              # used 3: want 5, not above 6 and 2 * 5 >= 6 -> keep 6
    assert hip_ops._async_attempts == 6
    collect(0, 30, 10)[1]()            # used 40: want 50
    assert hip_ops._async_attempts == 50
    p, c = collect(-5, 40, 10)         # sealed at 50 attempts: the next solve gets 100, and the error sticks to this one
    with pytest.raises(_lib.AsyncSolveTruncated):
        c()
    assert hip_ops._async_attempts == 100 and not hip_ops._pending_solves
    with pytest.raises(_lib.AsyncSolveTruncated):
        c()
    hip_ops._async_attempts = 200
    with pytest.raises(_lib.AsyncSolveTruncated):
        collect(-5, 150, 50)[1]()
    assert hip_ops._async_attempts == hip_ops.ASYNC_ATTEMPTS_MAX == 256
    collect(0, 1000, 0)[1]()           # capped
    assert hip_ops._async_attempts == 256


def test_last_stats_fills_itself_on_first_access():
    """ode_rl_amd.last_stats after an asynchronous solve (hip_ops.LazyStats): bound to the pending solve, it collects on the FIRST look
    of any kind -- and a failed solve raises there -- then behaves as the plain dict the synchronous path fills."""
    from ode_rl_amd import hip_ops

    class Pending:
        def __init__(self, result):
            self.result, self.collected = result, 0

        def collect(self):
            self.collected += 1
            if isinstance(self.result, BaseException):
                raise self.result
            return self.result, None

    for look in (lambda d: d["nfe"], lambda d: d.get("nfe"), lambda d: "nfe" in d, lambda d: len(d), lambda d: list(d), lambda d: d.keys(),
                 lambda d: d.items(), lambda d: d.values(), lambda d: d.copy(), lambda d: repr(d), lambda d: d == {}):
        d, p = hip_ops.LazyStats(), Pending({"nfe": 14, "n_accept": 2})
        d["stale"] = 1
        d._bind(p)
        assert p.collected == 0
        look(d)
        assert p.collected == 1 and dict.__getitem__(d, "nfe") == 14 and "stale" not in d
        look(d)
        assert p.collected == 1
    d, p = hip_ops.LazyStats(), Pending(AssertionError("underflow in dt"))
    d._bind(p)
    with pytest.raises(AssertionError, match="underflow"):
        d["nfe"]
    assert len(d) == 0 and p.collected == 1     # the error was reported once; the dict is empty afterwards
    d._bind(Pending({"nfe": 8}))
    d.clear()                                    # the synchronous path clears before it fills: a pending solve is dropped, not collected
    d.update({"nfe": 20})
    assert d["nfe"] == 20
