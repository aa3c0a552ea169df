"""Drop-in for the reference's `modules/DiffEqSolver.py` (`DiffEqSolver` :12-52, `ODEFunc` :57-80):
same constructor arguments, attribute names and return layouts, with `odeint` and the dynamics
evaluation executed by the HIP library instead of torchdiffeq + ATen."""
import torch
import torch.nn as nn

from ..helpers import utils
from ..odeint import odeint, conv_stack_of
from .. import hip_ops


class DiffEqSolver(nn.Module):
    def __init__(self, ode_func, method, odeint_rtol=1e-4, odeint_atol=1e-5, device=torch.device("cpu"), memory=False):
        super().__init__()
        self.ode_func = ode_func
        self.ode_method = method
        self.device = device
        self.memory = memory
        self.odeint_rtol = odeint_rtol
        self.odeint_atol = odeint_atol

    def forward(self, first_point, time_steps_to_predict, backwards=False):
        """Integrate z0 over `time_steps_to_predict`; returns (T,B,C,H,W) time-first, out[0] == z0."""
        if self.memory is True:
            # reference :30-42 -- one odeint call per output point with a 1-element t returns its
            # input, so h_next = 2*h_prev; batch-first result (degenerate, kept as is)
            y_is = [first_point]
            b, c, h, w = first_point.size()
            for i in range(len(time_steps_to_predict)):
                h_prev = y_is[-1]
                pred_m = odeint(self.ode_func, h_prev, time_steps_to_predict[i:i + 1], rtol=self.odeint_rtol,
                                atol=self.odeint_atol, method=self.ode_method)
                y_is.append(h_prev + pred_m.view(b, c, h, w))
            return torch.stack(y_is[1:]).permute(1, 0, 2, 3, 4)
        pred_y = odeint(self.ode_func, first_point, time_steps_to_predict, rtol=self.odeint_rtol,
                        atol=self.odeint_atol, method=self.ode_method)
        if pred_y.dim() == 3:
            pred_y = None  # reference :48-49
        return pred_y


class ODEFunc(nn.Module):
    def __init__(self, n_inputs=3, n_outputs=3, n_layers=4, n_units=256, downsize=False, nonlinear='relu',
                 final_act=True, net=None, device=torch.device("cpu")):
        super().__init__()
        self.device = device
        self.n_outputs = n_outputs
        if net is None:
            self.gradient_net = utils.create_convnet(n_inputs, n_outputs, n_layers, n_units, downsize, nonlinear,
                                                     final_act=final_act).to(device)
        else:
            self.gradient_net = net

    def forward(self, t_local, y, backwards=False):
        """dy/dt at y (t_local is ignored: the system is autonomous, reference :77)."""
        return hip_ops.convstack_forward(conv_stack_of(self), y, negate=backwards)
