#!/bin/bash
# training-step A/B: tests of the fused loss / SH concatenation, launch audit, ms/step with the fused loss and with the torch formulation
cd /root/repo
timeout 900 python -m pytest tests/test_train_loss.py tests/test_gpu_ops.py -x -q -m gpu -k "train_loss or fused_loss or sh_ or training" 2>&1 | tail -5
for m in palette nerf; do
  python3 profiles/train_launch_audit.py --model $m 2>&1 | grep -v "^/\|Warn\|_warn\|^\[W" > gpurun_out/audit_${m}.txt
  head -1 gpurun_out/audit_${m}.txt
  python3 profiles/train_step_bench.py --model $m --steps 100 2>&1 | tail -1
  [ -n "$TORCH_LOSS" ] && python3 profiles/train_step_bench.py --model $m --steps 100 --torch-loss 2>&1 | tail -1
done
