"""Synthetic Moving-MNIST batches rendered on the GPU.

Stands in for the reference's `MovingMNIST` dataset + `DataLoader` (/root/reference/dataloader.py:11-226) in its on-the-fly
mode (`frozen=False`, `is_train=True`): the same bouncing-digit walk and compositing, but the frames are produced by
`odehip_mmnist_render` directly in device memory, so end-to-end benchmarks and `train_batch` need neither the mp4/`.npy` files of
the "frozen" loader, nor cv2, nor a host->device copy per batch.  There is no MNIST file offline, so the default glyphs are
procedural 28x28 digits (`synthetic_digit_glyphs`); pass `glyphs=load_mnist(data_dir)` to use the real ones.

A batch is the dict the reference's collated loader yields ("observed_data", "data_to_predict" as (B,T,1,64,64) in
[-0.5, 0.5], "idx", "zeros"); `get_next_batch` mirrors `helpers/utils.py:101-147` for the keys ODEConvGRU consumes.
"""
import ctypes
import gzip
import os

import numpy as np
import torch

from . import _lib

_SEGMENTS = {  # seven-segment encoding a..g
    0: "abcdef", 1: "bc", 2: "abdeg", 3: "abcdg", 4: "bcfg", 5: "acdfg", 6: "acdefg", 7: "abc", 8: "abcdefg", 9: "abcdfg"}


def synthetic_digit_glyphs():
    """(10, 28, 28) uint8: seven-segment digits with soft edges (deterministic; stands in for MNIST)."""
    yy, xx = np.mgrid[0:28, 0:28].astype(np.float64)
    x0, x1, y0, y1, y2 = 8.0, 19.0, 4.0, 13.5, 23.0
    seg = {"a": (x0, y0, x1, y0), "b": (x1, y0, x1, y1), "c": (x1, y1, x1, y2), "d": (x0, y2, x1, y2),
           "e": (x0, y1, x0, y2), "f": (x0, y0, x0, y1), "g": (x0, y1, x1, y1)}
    out = np.zeros((10, 28, 28), dtype=np.uint8)
    for d, names in _SEGMENTS.items():
        img = np.zeros((28, 28))
        for n in names:
            ax, ay, bx, by = seg[n]
            t = np.clip(((xx - ax) * (bx - ax) + (yy - ay) * (by - ay)) / ((bx - ax) ** 2 + (by - ay) ** 2), 0.0, 1.0)
            dist = np.hypot(xx - (ax + t * (bx - ax)), yy - (ay + t * (by - ay)))
            img = np.maximum(img, np.clip(2.2 - dist, 0.0, 1.0))
        out[d] = np.round(img * 255.0).astype(np.uint8)
    return out


def load_mnist(data_dir):
    """The reference's `utils.load_mnist` (helpers/utils.py:60-66): (N, 28, 28) uint8 from train-images-idx3-ubyte.gz."""
    with gzip.open(os.path.join(data_dir, 'train-images-idx3-ubyte.gz'), 'rb') as f:
        return np.frombuffer(f.read(), np.uint8, offset=16).reshape(-1, 28, 28)


class MovingMNISTSynthetic:
    """Endless iterator of batches; `draw()` exposes the random initial state of a batch (for parity tests)."""

    def __init__(self, n_frames_input, n_frames_output, num_objects=(2,), batch_size=64, device=None, seed=0, glyphs=None):
        self.n_frames_input, self.n_frames_output = int(n_frames_input), int(n_frames_output)
        self.num_objects = list(num_objects)
        self.batch_size = int(batch_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MovingMNISTSynthetic renders with the HIP library: a GPU device is required (no CPU fallback)")
        self.rng = np.random.default_rng(seed)
        g = synthetic_digit_glyphs() if glyphs is None else np.ascontiguousarray(glyphs, dtype=np.uint8)
        if g.ndim != 3 or g.shape[1:] != (28, 28):
            raise ValueError("glyphs must be (N, 28, 28) uint8")
        self.glyphs_host = g
        self._glyphs = torch.from_numpy(g.copy()).to(self.device)
        # float32 arithmetic, as numpy evaluates `(images / 255.0) - 0.5` on the reference's float32 frames (dataloader.py:217-218)
        self._lut = torch.from_numpy((np.arange(256, dtype=np.float32) / 255.0) - 0.5).to(self.device)
        self._idx = 0

    def draw(self):
        """x, y, theta ~ U[0,1), U[0,1), U[0,2pi) and a glyph index per (sample, digit) -- dataloader.py:50-52, :91."""
        b, d = self.batch_size, int(self.rng.choice(self.num_objects))
        return {"x": self.rng.random((b, d)), "y": self.rng.random((b, d)), "theta": self.rng.random((b, d)) * 2 * np.pi,
                "ids": self.rng.integers(0, self.glyphs_host.shape[0], size=(b, d)).astype(np.int32)}

    def render(self, state):
        b, d = state["ids"].shape
        init = np.stack([state["x"], state["y"], np.cos(state["theta"]), np.sin(state["theta"])], axis=-1).astype(np.float64)
        if int(state["ids"].min()) < 0 or int(state["ids"].max()) >= self.glyphs_host.shape[0]:
            raise ValueError("digit id out of range")
        init_d = torch.from_numpy(np.ascontiguousarray(init)).to(self.device)
        ids_d = torch.from_numpy(np.ascontiguousarray(state["ids"], dtype=np.int32)).to(self.device)
        obs = torch.empty((b, self.n_frames_input, 1, 64, 64), device=self.device)
        pred = torch.empty((b, self.n_frames_output, 1, 64, 64), device=self.device)
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().odehip_mmnist_render(
                init_d.data_ptr(), ids_d.data_ptr(), self._glyphs.data_ptr(), self.glyphs_host.shape[0], self._lut.data_ptr(), b, d,
                self.n_frames_input, self.n_frames_output, obs.data_ptr() if self.n_frames_input else None,
                pred.data_ptr() if self.n_frames_output else None, stream))
        # the launch is asynchronous: keep the small operand tensors alive until the stream has consumed them
        init_d.record_stream(torch.cuda.current_stream())
        ids_d.record_stream(torch.cuda.current_stream())
        return obs, pred

    def __iter__(self):
        return self

    def __next__(self):
        obs, pred = self.render(self.draw())
        idx = torch.arange(self._idx, self._idx + self.batch_size)
        self._idx += self.batch_size
        return {"idx": idx, "observed_data": obs, "data_to_predict": pred, "zeros": torch.zeros(self.batch_size, 1, dtype=torch.float64)}


def get_next_batch(data_dict, opt=None):
    """`helpers/utils.py:101-147` for an ODEConvGRU batch: timesteps = arange(T_in + T_out) / (T_in + T_out) (float64) split into
    observed / to-predict; flow labels and masks are passed through when the loader provides them."""
    obs, pred = data_dict["observed_data"], data_dict["data_to_predict"]
    input_t, output_t = obs.size(1), pred.size(1)
    total_t = input_t + output_t
    ts = torch.tensor(np.arange(0, total_t) / total_t).to(obs.device)
    batch = {"observed_data": obs, "data_to_predict": pred, "timesteps": ts, "observed_tp": ts[:input_t], "tp_to_predict": ts[input_t:],
             "observed_mask": None, "mask_predicted_data": None, "in_flow_labels": None, "out_flow_labels": None}
    if data_dict.get("in_flow_labels") is not None:
        batch["in_flow_labels"] = data_dict["in_flow_labels"].to(obs.device)
        batch["out_flow_labels"] = data_dict["in_flow_labels"].to(obs.device)  # sic (helpers/utils.py:113)
    if data_dict.get("mask") is not None:
        batch["observed_mask"] = data_dict["mask"][:, :input_t].clone().to(obs.device)
        batch["mask_predicted_data"] = data_dict["mask"][:, input_t:].clone().to(obs.device)
    return batch
