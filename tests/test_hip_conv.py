"""GPU parity of the single-layer HIP conv (odehip_conv_q4) against torch CPU fp32 conv2d.
Tolerance: rel-L2 <= 2e-6 per layer (exact-fp32 MFMA vs oneDNN fp32; only summation order differs)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

CASES = [
    # (batch, cin1, cin2, cout, ks, relu)
    (3, 64, 0, 64, 3, True),
    (2, 128, 0, 64, 3, False),
    (1, 64, 0, 128, 3, True),
    (2, 32, 0, 32, 3, False),
    (5, 32, 32, 64, 5, False),
    (2, 64, 64, 128, 5, True),
    (2, 64, 0, 64, 1, True),
    (2, 64, 0, 128, 1, False),
    (1, 16, 0, 32, 3, False),
]


@pytest.mark.parametrize("b,cin1,cin2,cout,ks,relu", CASES)
def test_conv_matches_torch(cuda, b, cin1, cin2, cout, ks, relu):
    from ode_rl_amd import hip_ops
    g = torch.Generator().manual_seed(b * 1000 + cin1 + cout + ks)
    cin = cin1 + cin2
    x = torch.randn(b, cin, 16, 16, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=ks // 2)
    if relu:
        ref = torch.relu(ref)
    xd = x.to(cuda)
    src1 = hip_ops.nchw_to_q4(xd[:, :cin1].contiguous())
    src2 = hip_ops.nchw_to_q4(xd[:, cin1:].contiguous()) if cin2 else None
    wp = hip_ops.pack_conv_weight(w.to(cuda))
    out = hip_ops.q4_to_nchw(hip_ops.conv_q4(src1, wp, bias.to(cuda), cout, ks, src2=src2, relu=relu))
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) <= 2e-6


def test_layout_roundtrip(cuda):
    from ode_rl_amd import hip_ops
    x = torch.randn(3, 64, 16, 16, device=cuda)
    q = hip_ops.nchw_to_q4(x)
    assert q.shape == (3, 16, 256, 4)
    # Q4[b][c/4][p][c%4]
    ref = x.view(3, 16, 4, 256).permute(0, 1, 3, 2).contiguous()
    assert torch.equal(q, ref)
    assert torch.equal(hip_ops.q4_to_nchw(q), x)


def test_dgrad_weights_give_input_gradient(cuda):
    """pack(transpose_flip) turns the same kernel into the input-gradient conv."""
    from ode_rl_amd import hip_ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 16, 16, generator=g, requires_grad=True)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24.0
    gy = torch.randn(2, 32, 16, 16, generator=g)
    F.conv2d(x, w, None, padding=1).backward(gy)
    wp = hip_ops.pack_conv_weight(w.to(cuda), transpose_flip=True)
    out = hip_ops.q4_to_nchw(hip_ops.conv_q4(hip_ops.nchw_to_q4(gy.to(cuda)), wp, None, 64, 3))
    assert rel_l2(out, x.grad) <= 2e-6


def test_bad_shapes_raise(cuda):
    from ode_rl_amd import hip_ops
    x = hip_ops.nchw_to_q4(torch.randn(1, 64, 16, 16, device=cuda))
    wp = torch.zeros(48 * 64 * 9, device=cuda)
    with pytest.raises(ValueError):
        hip_ops.conv_q4(x, wp, None, 48, 3)  # cout not a multiple of 32
    with pytest.raises(ValueError):
        hip_ops.conv_q4(x, wp, None, 64, 7)  # unsupported kernel size
    with pytest.raises(RuntimeError):
        hip_ops.nchw_to_q4(torch.randn(1, 64, 16, 16))  # CPU tensor: no fallback


@pytest.mark.parametrize("b,cin,cout,relu", [(3, 64, 64, True), (2, 32, 32, False), (1, 16, 64, False), (2, 128, 64, True),
                                            (5, 64, 128, False)])
def test_winograd_conv_matches_torch(cuda, b, cin, cout, relu):
    """Winograd F(2x2,3x3) kernel: still fp32 arithmetic, reassociated; rel-L2 <= 5e-6 vs torch CPU conv2d."""
    from ode_rl_amd import hip_ops
    g = torch.Generator().manual_seed(b * 77 + cin + cout)
    x = torch.randn(b, cin, 16, 16, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    if relu:
        ref = torch.relu(ref)
    wd = w.to(cuda)
    out = hip_ops.q4_to_nchw(hip_ops.conv_q4(hip_ops.nchw_to_q4(x.to(cuda)), hip_ops.pack_conv_weight(wd), bias.to(cuda), cout, 3,
                                             relu=relu, w_wino=hip_ops.pack_conv_weight_winograd(wd)))
    assert rel_l2(out, ref) <= 5e-6
    if cin % 32 != 0:
        return
    # dgrad form (a conv with cin and cout swapped)
    gy = torch.randn(b, cout, 16, 16, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=1).backward(gy)
    dg = hip_ops.q4_to_nchw(hip_ops.conv_q4(hip_ops.nchw_to_q4(gy.to(cuda)), hip_ops.pack_conv_weight(wd, transpose_flip=True), None,
                                            cin, 3, w_wino=hip_ops.pack_conv_weight_winograd(wd, transpose_flip=True)))
    assert rel_l2(dg, xr.grad) <= 5e-6


@pytest.mark.parametrize("b,cin1,cin2,cout,relu", [(3, 64, 64, 128, False), (2, 64, 64, 64, True), (1, 8, 0, 32, False),
                                                   (5, 32, 32, 64, False), (2, 128, 128, 256, False), (70, 64, 64, 128, False),
                                                   (4, 64, 64, 128, False), (4, 64, 64, 64, True), (16, 64, 64, 64, False), (9, 64, 64, 128, False)])
def test_winograd5_conv_matches_torch(cuda, b, cin1, cin2, cout, relu):
    """Winograd F(2x2,5x5) kernel (csrc/conv_wino5.hip; the ConvGRU's convolutions, two sources = torch.cat(x, h) without a copy):
    fp32 arithmetic, reassociated through the 6-point transforms: 2.6e-6 rel-L2 per layer on the CPU model of it
    (tools/experiments/winograd_f25_error.py); tolerance 1e-5 against torch's CPU conv2d, and the input-gradient form.
    Round 4: small batches SPLIT the chunk chain over 2 / 4 / 8 workgroups per output tile (the reference's batch 4: 8 and 8; the cases
    here reach every split count and the unsplit kernel); the partials are added in a fixed order: a second launch is bitwise equal."""
    from ode_rl_amd import hip_ops
    g = torch.Generator().manual_seed(b * 131 + cin1 + cout)
    cin = cin1 + cin2
    x = torch.randn(b, cin, 16, 16, generator=g)
    w = (torch.rand(cout, cin, 5, 5, generator=g) * 2 - 1) / (cin * 25) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=2)
    if relu:
        ref = torch.relu(ref)
    wd = w.to(cuda)
    src1 = hip_ops.nchw_to_q4(x[:, :cin1].contiguous().to(cuda))
    src2 = hip_ops.nchw_to_q4(x[:, cin1:].contiguous().to(cuda)) if cin2 else None
    out = hip_ops.q4_to_nchw(hip_ops.conv_q4(src1, hip_ops.pack_conv_weight(wd), bias.to(cuda), cout, 5, src2=src2, relu=relu,
                                             w_wino=hip_ops.pack_conv_weight_winograd5(wd)))
    direct = hip_ops.q4_to_nchw(hip_ops.conv_q4(src1, hip_ops.pack_conv_weight(wd), bias.to(cuda), cout, 5, src2=src2, relu=relu))
    assert rel_l2(direct, ref) <= 2e-6
    assert not torch.equal(out, direct)          # the Winograd kernel ran (it is not bit-identical to the direct one)
    assert rel_l2(out, ref) <= 1e-5
    for _ in range(3):                           # deterministic (also across the split launches' last-arriver reduction)
        again = hip_ops.q4_to_nchw(hip_ops.conv_q4(src1, hip_ops.pack_conv_weight(wd), bias.to(cuda), cout, 5, src2=src2, relu=relu,
                                                   w_wino=hip_ops.pack_conv_weight_winograd5(wd)))
        assert torch.equal(again, out)
    if cin % 32 != 0 or b > 8:
        return
    gy = torch.randn(b, cout, 16, 16, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=2).backward(gy)
    dg = hip_ops.q4_to_nchw(hip_ops.conv_q4(hip_ops.nchw_to_q4(gy.to(cuda)), hip_ops.pack_conv_weight(wd, transpose_flip=True), None,
                                            cin, 5, w_wino=hip_ops.pack_conv_weight_winograd5(wd, transpose_flip=True)))
    assert rel_l2(dg, xr.grad) <= 1e-5


@pytest.mark.gpu
def test_pack_many_equals_the_single_packs(cuda):
    """odehip_pack_conv_weights (one launch for a whole conv stack) writes exactly what the per-tensor pack calls write"""
    from ode_rl_amd import hip_ops
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(64, 64, 3, 3, generator=g), torch.randn(128, 64, 3, 3, generator=g), torch.randn(64, 128, 3, 3, generator=g),
          torch.randn(32, 64, 5, 5, generator=g), torch.randn(64, 32, 1, 1, generator=g)]
    ws = [w.to(cuda) for w in ws]
    jobs = []
    for w in ws:
        for tf in (False, True):
            jobs.append((w, 0, tf))
            if w.shape[-1] == 3:
                jobs.append((w, 1, tf))
            if w.shape[-1] == 5:
                jobs.append((w, 2, tf))
    jobs = jobs * 3   # 48 jobs: more than one launch of ODEHIP_MAX_PACK_JOBS
    got = hip_ops.pack_conv_weights_many(jobs)
    single = (hip_ops.pack_conv_weight, hip_ops.pack_conv_weight_winograd, hip_ops.pack_conv_weight_winograd5)
    for (w, kind, tf), out in zip(jobs, got):
        assert torch.equal(out, single[kind](w, transpose_flip=tf)), (tuple(w.shape), kind, tf)
