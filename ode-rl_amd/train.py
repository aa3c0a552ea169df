"""Training-step harness around the HIP hot path: the ODEConvGRU branch of the reference's
`train_batch` (train_test.py:169-207) and its checkpoint format (helpers/utils.py:212-252), without the per-step host copies
of the reference's loop (train_test.py:50 `pred.detach().cpu()`) and with anomaly detection off (train_test.py:5)."""
import os
import pickle

import torch


def train_batch(model, batch_dict, optimizer, async_solver=False):
    """One optimisation step.  batch_dict: 'observed_data' (B,T_in,C,H,W) and 'data_to_predict' (B,T_out,C,H,W) in [-0.5, 0.5] as the
    reference's loaders deliver them, plus 'observed_tp' / 'tp_to_predict'.  Returns (pred * 255, truth * 255, loss tensor, loss_dict)
    -- the loss stays on the device (no .item() synchronisation here).

    async_solver (opt-in): a dopri5 solve inside the model only enqueues its attempted steps and the backward pass collects its outcome
    (ode_rl_amd.set_async_dopri5): the host goes on enqueueing decoder, loss and backward while the device still integrates.  A solver
    error (dt underflow, non-finite state) then surfaces at loss.backward(), not inside the forward as in torchdiffeq.  A solve that
    needs more attempted steps than were enqueued up front is SEALED on the device (its unreached frames are NaN, never stale memory)
    and reported as AsyncSolveTruncated by the backward pass: nothing has touched the parameters at that point, so the step is
    repeated here on the synchronous path, from the module buffers (BatchNorm statistics) of before the sealed pass (and the next
    asynchronous solve enqueues more attempts)."""
    dev = next(model.parameters()).device
    inp = batch_dict["observed_data"].to(dev) + 0.5          # train_test.py:180: [-0.5, 0.5] -> [0, 1]
    out = batch_dict["data_to_predict"].to(dev) + 0.5
    from . import _lib, hip_ops

    def step():
        optimizer.zero_grad()   # torch 2 default (set_to_none=True), as train_test.py:176 today: no fill + accumulate kernels per parameter
        pred = model.get_prediction(inp, batch_dict=batch_dict)
        loss = model.get_loss(pred, out)
        loss.backward()
        return pred, loss

    if async_solver:
        # what a sealed pass could leave behind besides gradients (zeroed by the repeat): module buffers -- a BatchNorm behind the
        # solver (VidODE's flow decoder) would fold the NaN frames into its running statistics
        buffers = [(b, b.detach().clone()) for b in model.buffers()]
        was = hip_ops.set_async_dopri5(True)
        try:
            try:
                pred, loss = step()
                hip_ops.collect_pending_solves()   # a solve nobody differentiated through (none in the models here) is checked too
            finally:
                hip_ops.set_async_dopri5(was)
        except _lib.AsyncSolveTruncated:
            with torch.no_grad():
                for b, kept in buffers:
                    b.copy_(kept)
            pred, loss = step()
    else:
        pred, loss = step()
    optimizer.step()
    return pred.detach() * 255.0, out * 255.0, loss.detach(), {"Per Step Loss": loss.detach()}


def checkpoint_name(ckpt_id, step):
    return f"{ckpt_id}_{step:010d}.pickle"                    # helpers/utils.py:215-217


def save_model_params(model, optimizer, epoch, step, logdir, ckpt_id):
    """Same pickle as helpers/utils.py:212-226: {'epoch', 'step', 'state_dict', 'optimizer'} (aliased state_dict keys kept)."""
    path = os.path.join(logdir, checkpoint_name(ckpt_id, step))
    blob = {"epoch": epoch, "step": step,
            "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "optimizer": optimizer.state_dict()}
    with open(path, "wb") as fh:
        pickle.dump(blob, fh, protocol=pickle.HIGHEST_PROTOCOL)
    return path


def load_model_params(model, path, optimizer=None):
    with open(path, "rb") as fh:
        blob = pickle.load(fh)
    model.load_state_dict(blob["state_dict"])
    if optimizer is not None and "optimizer" in blob:
        optimizer.load_state_dict(blob["optimizer"])
    return blob.get("epoch"), blob.get("step")
