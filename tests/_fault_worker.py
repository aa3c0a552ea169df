"""Subprocess of tests/test_hip_errors.py::test_a_lost_partner_*: runs ONE workload with ODEHIP_FAULT_INJECT=1 (logical workgroup 0 of the
sixteen-workgroup walk leaves in front of row 1, its partners' capped waits give up) and prints what a caller would see."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    which = sys.argv[1]
    import ode_rl_amd
    from ode_rl_amd import _lib
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    rec = {"which": which}
    if which == "trajectory":   # a whole fixed-grid trajectory at batch 4: host-side guard launch behind the walk
        f = ode_rl_amd.ODEFunc(64, 64, 3, 64, False, "relu", final_act=False).to(dev)
        z0 = torch.randn(4, 64, 16, 16, device=dev) * 0.5
        t = torch.arange(4, 8, dtype=torch.float64) / 8
        with torch.no_grad():
            good = ode_rl_amd.odeint(f, z0, t, method="rk4")
            torch.cuda.synchronize()
            os.environ["ODEHIP_FAULT_INJECT"] = "1"
            try:
                out = ode_rl_amd.odeint(f, z0, t, method="rk4")
                torch.cuda.synchronize()
                rec["raised_in_call"] = False
            except (_lib.OdeHipError, ValueError) as e:
                out, rec["raised_in_call"] = None, True
                rec["message"] = str(e)[:160]
            os.environ.pop("ODEHIP_FAULT_INJECT")
            if out is not None:
                rec["first_frame_is_z0"] = bool(torch.equal(out[0], z0))
                rec["later_frames_all_nan"] = bool(torch.isnan(out[1:]).all())
            try:
                again = ode_rl_amd.odeint(f, z0, t, method="rk4")   # the sticky word: this (or the failed call itself) must raise
                torch.cuda.synchronize()
                rec["next_call_raised"] = False
                rec["next_call_equals_good"] = bool(torch.equal(again, good))
            except (_lib.OdeHipError, ValueError) as e:
                rec["next_call_raised"] = True
                rec["message"] = str(e)[:160]
            # after the error the library runs one launch per layer and is usable again
            after = ode_rl_amd.odeint(f, z0, t, method="rk4")
            rec["usable_afterwards"] = bool(torch.equal(after, good))
    else:   # the encoder loop at batch 2: its Euler steps are single-evaluation walks WITHOUT a guard launch (in-kernel NaN fill)
        import argparse
        from ode_rl_amd.modules.DiffEqSolver import ODEFunc
        from ode_rl_amd.modules.ODEConvGRUCell import ODEConvGRUCell
        fe = ODEFunc(64, 64, 3, 64, False, "relu", final_act=False)
        opt = argparse.Namespace(input_size=16, total_len=8, batch_size=2, z_sample=False)
        cell = ODEConvGRUCell(fe, opt, (16, 16), 64, out_ch=64, device=dev).to(dev)
        x = torch.randn(4, 2, 64, 16, 16, device=dev)
        ts = torch.arange(4, dtype=torch.float64) / 8
        with torch.no_grad():
            # the hidden states (run_ode_conv_gru's latent_ys), not forward()'s (mean, std): the 1x1 head behind them ends in ReLUs, and
            # this library's ReLU is v_max_f32(x, 0), which maps NaN to 0 (torch.relu would propagate it)
            good, _ = cell.run_ode_conv_gru(x, ts)
            torch.cuda.synchronize()
            os.environ["ODEHIP_FAULT_INJECT"] = "1"
            try:
                mean, _ = cell.run_ode_conv_gru(x, ts)
                torch.cuda.synchronize()
                rec["raised_in_call"] = False
                rec["output_has_nan"] = bool(torch.isnan(mean).any())
                rec["output_equals_good"] = bool(torch.equal(mean, good))
            except (_lib.OdeHipError, ValueError) as e:
                rec["raised_in_call"] = True
                rec["message"] = str(e)[:160]
            os.environ.pop("ODEHIP_FAULT_INJECT")
            try:
                cell.run_ode_conv_gru(x, ts)
                torch.cuda.synchronize()
                rec["next_call_raised"] = False
            except (_lib.OdeHipError, ValueError) as e:
                rec["next_call_raised"] = True
                rec["message"] = str(e)[:160]
            after, _ = cell.run_ode_conv_gru(x, ts)
            rec["usable_afterwards"] = bool(torch.equal(after, good)) or bool(torch.allclose(after, good, rtol=1e-5, atol=1e-6))
    print("FAULT_RECORD " + json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
