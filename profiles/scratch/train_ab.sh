#!/bin/bash
# training-step A/B: training-side tests, launch audit, ms/step with the fused loss (TORCH_LOSS=1: and with the torch formulation)
cd /root/repo
timeout 1500 python -m pytest tests/test_train_loss.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "train or loss or sh_ or mlp or heads or shade or dropin" 2>&1 | tail -4
for m in palette nerf; do
  python3 profiles/train_launch_audit.py --model $m 2>&1 | grep -v "^/\|Warn\|_warn\|^\[W" > gpurun_out/audit_${m}.txt
  head -1 gpurun_out/audit_${m}.txt
  python3 profiles/train_step_bench.py --model $m --steps 100 2>&1 | tail -1
  if [ -n "$TORCH_LOSS" ]; then python3 profiles/train_step_bench.py --model $m --steps 100 --torch-loss 2>&1 | tail -1; fi
done
true
