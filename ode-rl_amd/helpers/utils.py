"""The one helper of the reference's `helpers/utils.py` that sits on the hot path: `create_convnet`
(/root/reference/helpers/utils.py:158-183).  Same signature, same nn.Sequential layout (conv layers at
indices 0, 2, 4, ... so `state_dict` keys match existing checkpoints)."""
import torch.nn as nn


def create_convnet(n_inputs, n_outputs, n_layers=1, n_units=128, downsize=False, nonlinear='tanh', final_act=True):
    if nonlinear == 'tanh':
        act = nn.Tanh
    elif nonlinear == 'relu':
        act = nn.ReLU
    else:
        raise NotImplementedError('Wrong activation function')
    hidden = (lambda: nn.Conv2d(n_units, n_units, 4, 2, 1)) if downsize else (lambda: nn.Conv2d(n_units, n_units, 3, 1, 1))
    layers = [nn.Conv2d(n_inputs, n_units, 3, 1, 1)]
    for _ in range(n_layers):
        layers += [act(), hidden()]
    layers += [act(), nn.Conv2d(n_units, n_outputs, 3, 1, 1)]
    if final_act is True:
        layers.append(nn.Tanh())
    return nn.Sequential(*layers)
