#!/bin/bash
# The GPU suite with every fast path switched off (one launch per layer, direct 5x5 kernel unsplit, direct weight gradients, the library's
# codec backward, the four-workgroup walk at small batches, re-integrating dopri5 backward, host-driven adjoint, per-layer Euler steps,
# module-by-module flow decoder): the slow paths are the A/B references of the fast ones and must pass the same parity tests.
# Run on the GPU box from the repo root; tests of the switched-off kernels themselves skip.
export ODEHIP_PERSISTENT=0 ODEHIP_WINO5=0 ODEHIP_WGRAD_WINO=0 ODEHIP_WGRAD_WINO5=0 ODEHIP_CODEC_BACKWARD=0 ODEHIP_PERSIST16=0 ODEHIP_DOPRI5_SAVE=0 \
       ODEHIP_ADJOINT_DEVICE=0 ODEHIP_WINO5_SPLIT=0 ODEHIP_EVAL_WALK=0 ODEHIP_FLOW_FUSED=0
exec python -m pytest tests -m gpu -q -p no:cacheprovider "$@"
