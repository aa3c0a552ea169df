"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's on-the-fly Moving-MNIST generator,
used to check `odehip_mmnist_render`.  Follows /root/reference/dataloader.py:
  * trajectory()  <- MovingMNIST.get_random_trajectory (:47-80), with the three random draws (x, y, theta) passed in;
  * render()      <- MovingMNIST.generate_moving_mnist (:82-103) + the normalisation of __getitem__ (:217-218).
Parity status: the reference holds no fixture for its generator (its draws come from Python's global `random`), so this
restatement is pinned by hand-checked cases in tests/test_mmnist.py only -- "parity unpinned" beyond those."""
import numpy as np

IMAGE, DIGIT, STEP = 64, 28, 0.1


def trajectory(x, y, theta, seq_length):
    canvas_size = IMAGE - DIGIT
    v_y = np.sin(theta)
    v_x = np.cos(theta)
    start_y = np.zeros(seq_length)
    start_x = np.zeros(seq_length)
    for i in range(seq_length):
        y += v_y * STEP
        x += v_x * STEP
        if x <= 0:
            x = 0
            v_x = -v_x
        if x >= 1.0:
            x = 1.0
            v_x = -v_x
        if y <= 0:
            y = 0
            v_y = -v_y
        if y >= 1.0:
            y = 1.0
            v_y = -v_y
        start_y[i] = y
        start_x[i] = x
    return (canvas_size * start_y).astype(np.int32), (canvas_size * start_x).astype(np.int32)


def render(glyphs, digit_ids, xs, ys, thetas, n_frames_input, n_frames_output):
    """One sample: digit d = glyphs[digit_ids[d]] walks from (xs[d], ys[d]) with heading thetas[d].
    Returns (observed (T_in,1,64,64), to_predict (T_out,1,64,64)) float32 in [-0.5, 0.5]."""
    total = n_frames_input + n_frames_output
    data = np.zeros((total, IMAGE, IMAGE), dtype=np.float32)
    for d in range(len(digit_ids)):
        start_y, start_x = trajectory(float(xs[d]), float(ys[d]), float(thetas[d]), total)
        digit_image = glyphs[digit_ids[d]]
        for i in range(total):
            top, left = start_y[i], start_x[i]
            data[i, top:top + DIGIT, left:left + DIGIT] = np.maximum(data[i, top:top + DIGIT, left:left + DIGIT], digit_image)
    images = data[:, None, :, :]
    out = (images / 255.0) - 0.5  # float32 frames / python scalars: numpy stays in float32, then .float() (:217-218)
    assert out.dtype == np.float32
    return out[:n_frames_input], out[n_frames_input:]
