"""Drop-in for the reference's `modules/ConvGRUCell.py:11-86` (the cell only; `ConvEncoder` at :88-120 is not on the
hot path).  Same constructor, same parameter layout (`conv_gates.{0,1}`, `conv_can.{0,1}`), forward on the HIP library."""
import torch
import torch.nn as nn

from .. import hip_ops


class ConvGRUCell(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias=True, dtype=None, padding=None):
        super().__init__()
        self.height, self.width = input_size
        self.input_channels = input_dim
        self.hidden_dim = hidden_dim
        self.padding = (kernel_size - 1) // 2 if padding is None else padding
        self.bias = bias
        self.dtype = dtype if dtype is not None else torch.FloatTensor
        # update (z) and reset (r) gates in one conv: z = first hidden_dim channels, r = second (reference :75-77)
        self.conv_gates = nn.Sequential(
            nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, kernel_size, 1, self.padding),
            nn.GroupNorm(2 * hidden_dim // 32, 2 * hidden_dim))
        self.conv_can = nn.Sequential(
            nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, 1, self.padding),
            nn.GroupNorm(hidden_dim // 32, hidden_dim))

    def init_hidden(self, batch_size):
        return torch.zeros(batch_size, self.hidden_dim, self.height, self.width, device=self.conv_can[0].weight.device)

    def _packed(self):
        p = getattr(self, "_hip_cell", None)
        if p is None:
            p = hip_ops.PackedCell(self)
            object.__setattr__(self, "_hip_cell", p)
        return p

    def forward(self, input_tensor=None, h_cur=None, seq_len=10, mask=None, dim=0):
        """input_tensor (seq_len, b, input_dim, h, w); returns (stack of h over the sequence, last h).
        `mask` is accepted and ignored, as in the reference (:55-86)."""
        dev = self.conv_can[0].weight.device
        if h_cur is None:
            h_cur = torch.zeros(input_tensor.size(1), self.hidden_dim, self.height, self.width, device=dev)
        outs = []
        for index in range(seq_len):
            if input_tensor is None:
                x = torch.zeros(h_cur.size(0), self.input_channels, self.height, self.width, device=dev)
            else:
                x = input_tensor[index, ...]
            if torch.is_grad_enabled() and (x.requires_grad or h_cur.requires_grad or any(p.requires_grad for p in self.parameters())):
                from ..autograd import cell_with_grad
                h_cur = cell_with_grad(self._packed(), x, h_cur)
            else:
                h_cur = hip_ops.convgru_cell_forward(self._packed(), x, h_cur)
            outs.append(h_cur)
        return torch.stack(outs, dim=dim), h_cur
