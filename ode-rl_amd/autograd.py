"""Autograd glue for odeint (backward passes).  Round-1 state: forward only."""


def odeint_with_grad(func, y0, t, rtol, atol, method):
    raise NotImplementedError(
        "odeint(HIP): backward is not implemented yet; call under torch.no_grad() or detach the inputs")
